// Mask priors of Trainer.make_loss_ops (cub/code/SB_model48i/model.py:652-797) fused into one streaming
// pass per view (forward sums) and one fused backward that emits d(loss)/d(logits) directly.
//
//  view 0 : categorical KL (model.py:21-25,659-665), entropy / cross-entropy (model.py:667-681),
//           Mumford-Shah + area (model.py:744-769, nn.py:1366-1398), patch (model.py:771-783),
//           improper GMRF on the noise-free logits (nn.py:1444-1451)
//  view 1 : categorical KL, variance (model.py:683-719; moments from ups_spatial_moments)
//
// Forward: GP = pow2 >= P adjacent lanes own the parts of one pixel (part reductions = shuffles), consecutive lane groups
// own consecutive pixels.  Backward: a block stages a tile of consecutive pixels of every map it reads in LDS (16-byte
// accesses, tile.h; the maps with finite-difference stencils come with the rows above and below), ONE THREAD OWNS ONE
// PIXEL and walks its P parts in LDS (odd pixel pitch: conflict-free, no cross-lane traffic -- the lane-per-part form spent
// 24 ds_bpermute per 4 pixels on the six part reductions), and the result tiles go back out the same way.
//
// sums[16] (written by the forward finalize):
//   0 sum m*log(P*m+1e-20)   1 sum_pix CE/entropy   2 sum hard*(1-rect)   3 sum 0.5*(dy^2+dx^2)
//   4 sum_np R^2   5 sum_np S^2   6 sum_np Rsmooth^2   7 sum_np Rcontour^2
// per_np (view 0) [n][P][8]: 0 S = sum m, 1 R = sum r, 2 Rsmooth, 3 Rcontour
// per_np (view 1) = stats of ups_spatial_moments: 0 max, 1 Z, 2 S0, 3 Sy, 4 Sx, 5 Q
#include "common.h"
#include "tile.h"

namespace {

constexpr int NSLAB = 64;     // row slabs per image in the forward pass (n x NSLAB blocks: 4 096 at B = 64, two tiles each at 128x128)

struct PriorK {
    int n, h, w, P, view, entropy_ce, half_h, half_w, variant;
    float gamma, ms_alpha, ms_lambda, w_kl, w_entropy, w_ms, w_area, w_patch, w_gmrf, w_var, w_msl;
    const float* l; const float* l_mean; const float* m; const float* hard; const int* px;
    float* per_np; float* sums; const float* g_hard; float* dl; float* ws; float* dl_rec;
};

// Logical block index for a 1-D grid whose consecutive logical blocks share halo rows: hardware block b runs on XCD b % 8, so
// logical = (b % 8) * (total / 8) + b / 8 puts a run of consecutive logical blocks on ONE XCD (its L2 then serves the rows that
// neighbouring tiles re-read; with the identity order every halo row came from HBM / Infinity Cache once per XCD that needed it).
__device__ __forceinline__ int xcd_logical_block() {
    const int total = gridDim.x, b = blockIdx.x;
    return (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
}

template <int GP>
__device__ inline float gsum(float v) {
#pragma unroll
    for (int o = GP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, GP);
    return v;
}
template <int GP>
__device__ inline float gmax(float v) {
#pragma unroll
    for (int o = GP / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, GP));
    return v;
}

__device__ inline float mval(const float* m, long long img_base, int y, int x, int h, int w, int P, int c) {
    return ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? m[(img_base + (long long)y * w + x) * P + c] : 0.f;
}

// value of a staged map at the pixel `dq` pixels after staged pixel hp (0 outside the image: SAME-padded differences)
__device__ inline float tval(const float* t, int PP, int hp, int dq, int c, int y, int x, int h, int w) {
    return ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? t[(hp + dq) * PP + c] : 0.f;
}

// ws layout: glob_partial[n][NSLAB][4], then np_partial[n][NSLAB][P][4]
// One block = one slab of rows of one image, walked in tiles of tpx (<= 256) pixels.  Per tile: stage m (+ the row below) and,
// for view 0, l_mean (same), l and hard; phase 1: one thread per pixel walks its parts (log-soft-max, entropy / CE, patch,
// GMRF and the Mumford-Shah term r, which it leaves in the hard slot with the sign marking the contour branch); phase 2:
// threads = (part, sub-lane) sum the columns of the staged tile into the per-part accumulators S, R, Rsmooth, Rcontour.
// PC / LW: the part count and log2 of the image width as compile-time constants for the common shapes (0 / -1: run-time values).
// (__logf / __expf / __fdividef: the hardware's log2 / exp2 / rcp, ~1e-7 relative -- two orders below the 1e-3 parity bar; the
// libm forms are 15-20 VALU instructions each.)
// These launches are VALU-bound (SQ counters, profiles/round4_pmc_part_kernels.txt: the vector ALU busy 60-80 % of the CU-busy
// cycles): the item index -> (pixel, part) and pixel -> (row, column) divisions by run-time values were a third of the instructions.
template <int PC, int LW>
__global__ __launch_bounds__(256, 3) void prior_fwd_kernel(const PriorK p, int rows_per_slab, int tpx) {
    extern __shared__ __attribute__((aligned(16))) float ts[];
    const int P = PC > 0 ? PC : p.P, PP = tile_pitch(P);
    const int W = LW >= 0 ? (1 << (LW >= 0 ? LW : 0)) : p.w;
    const int halo = p.view == 0 ? W + 1 : 0;
    float* tm = ts;
    float* tlm = tm + (size_t)(tpx + halo) * PP;
    float* tl = tlm + (p.view == 0 ? (size_t)(tpx + halo) * PP : 0);
    float* th = tl + (p.view == 0 ? (size_t)tpx * PP : 0);
    float* scratch = th + (p.view == 0 ? (size_t)tpx * PP : 0);      // red_np[NS][P][4], red4[4]
    const int NS = 256 / P;                                           // sub-lanes per part in phase 2
    float* red4 = scratch + (size_t)NS * P * 4;
    int* cpx = (int*)(red4 + 4);                                      // rectangle centres of this image [P][2]
    float* pst = (float*)(cpx + 2 * P);                               // per-pixel log-sum-exp of the tile [tpx]
    const int lb = xcd_logical_block();
    const int n = lb / NSLAB, slab = lb - n * NSLAB;
    if (p.view == 0 && p.px)
        for (int i = threadIdx.x; i < 2 * P; i += 256) cpx[i] = p.px[(long long)n * P * 2 + i];
    const int y0 = slab * rows_per_slab, y1 = max(y0, min(p.h, y0 + rows_per_slab));
    const int hw = p.h * W;
    const long long img = (long long)n * hw;
    float kl = 0.f, ent = 0.f, patch = 0.f, gmrf = 0.f;
    float S = 0.f, R = 0.f, Rs = 0.f, Rc = 0.f;
    const int c2 = threadIdx.x % P, s2 = threadIdx.x / P;            // phase-2 role
    const int q0 = y0 * W, q1 = y1 * W;
    // software pipeline over the slab's tiles: all maps of a tile are requested at once (one HBM round trip, tile.h) and the NEXT
    // tile's pieces are in flight while the current one is computed
    TileReq<3> rm, rlm;         // (P <= 11 at 128-wide images: everything in one round trip; larger P: the rest synchronously)
    TileReq<2> rl, rh;
    auto request = [&](int t0) {
        const int cnt = min(tpx, q1 - t0), cnt_h = min(cnt + halo, hw - t0);
        tile_request(rm, p.m + (img + t0) * P, cnt_h, P);
        if (p.view == 0) {
            tile_request(rlm, p.l_mean + (img + t0) * P, cnt_h, P);
            tile_request(rl, p.l + (img + t0) * P, cnt, P);
            tile_request(rh, p.hard + (img + t0) * P, cnt, P);
        }
    };
    if (q0 < q1) request(q0);
    for (int t0 = q0; t0 < q1; t0 += tpx) {
        const int cnt = min(tpx, q1 - t0);
        const int cnt_h = min(cnt + halo, hw - t0);
        __syncthreads();
        tile_commit(rm, p.m + (img + t0) * P, cnt_h, P, PP, tm);
        if (p.view == 0) {
            tile_commit(rlm, p.l_mean + (img + t0) * P, cnt_h, P, PP, tlm);
            tile_commit(rl, p.l + (img + t0) * P, cnt, P, PP, tl);
            tile_commit(rh, p.hard + (img + t0) * P, cnt, P, PP, th);
        }
        if (t0 + tpx < q1) request(t0 + tpx);
        __syncthreads();
        // element-wise work on ITEMS (pixel, part) over all 256 threads; only the log-sum-exp stays per pixel (see prior_bwd_kernel)
        if (p.view == 0) {
            for (int px = threadIdx.x; px < cnt; px += 256) {
                const float* lrow = tl + px * PP;
                float mx = -INFINITY;
                for (int c = 0; c < P; ++c) mx = fmaxf(mx, lrow[c]);
                float se = 0.f;
                for (int c = 0; c < P; ++c) se += __expf(lrow[c] - mx);
                pst[px] = mx + __logf(se);
            }
            __syncthreads();
        }
        // (items over all 256 threads -- exactly cnt * P / 256 each; the (part, pixel lane) form of the backward kernel measured
        // 19 % slower here: 250 busy threads with 5 or 6 pixels each)
        const int items = cnt * P;
        for (int it = threadIdx.x; it < items; it += 256) {
            const int px = it / P, c = it - px * P;
            const float* tmc = tm + px * PP + c;
            const float mc = tmc[0];
            kl += mc * __logf((float)P * mc + 1e-20f);
            if (p.view == 0) {
                const int q = t0 + px;
                const int yy = q / W, xx = q - yy * W;
                const bool vr = xx + 1 < W, vd = yy + 1 < p.h;
                const float sl = tl[px * PP + c] - pst[px];
                const float hv = th[px * PP + c];
                ent += -(p.entropy_ce ? hv : mc) * sl;
                const float* tlc = tlm + px * PP + c;
                const float lm = tlc[0];
                const float lr = vr ? tlc[PP] : 0.f, ld = vd ? tlc[W * PP] : 0.f;
                if (p.variant == 0) {
                    const bool in_rect = abs(yy - cpx[2 * c]) <= p.half_h && abs(xx - cpx[2 * c + 1]) <= p.half_w;
                    patch += hv * (in_rect ? 0.f : 1.f);
                } else {
                    // SB_model48c: Mumford-Shah on the noise-free logits, min(alpha * g, lambda) summed (patch slot)
                    const float gw = 0.25f * (lm - lr), gh = 0.25f * (lm - ld);
                    patch += fminf(p.ms_alpha * (gw * gw + gh * gh), p.ms_lambda);
                }
                if (vd) { const float d = ld - lm; gmrf += 0.5f * d * d; }
                if (vr) { const float d = lr - lm; gmrf += 0.5f * d * d; }
                const float mr = vr ? tmc[PP] : 0.f, md = vd ? tmc[W * PP] : 0.f;
                const float gw = 0.25f * (mc - mr), gh = 0.25f * (mc - md);
                const float g = p.ms_alpha * (gw * gw + gh * gh);
                const float r = fminf(g, p.ms_lambda);
                th[px * PP + c] = (g < p.ms_lambda) ? r : -r;            // sign = contour branch (r = lambda > 0 there)
            }
        }
        if (p.view == 0) {
            __syncthreads();
            if (s2 < NS) {
                for (int px = s2; px < cnt; px += NS) {
                    S += tm[px * PP + c2];
                    const float r = th[px * PP + c2];
                    const float ar = fabsf(r);
                    R += ar;
                    if (__float_as_uint(r) >> 31) Rc += ar; else Rs += ar;
                }
            }
        }
    }
    // per-(n,p) partials
    __syncthreads();
    if (s2 < NS) {
        float* d = scratch + ((size_t)s2 * P + c2) * 4;
        d[0] = S; d[1] = R; d[2] = Rs; d[3] = Rc;
    }
    __syncthreads();
    float* np_partial = p.ws + (long long)p.n * NSLAB * 4;
    if (threadIdx.x < P) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < NS; ++q)
            for (int k = 0; k < 4; ++k) o[k] += scratch[((size_t)q * P + threadIdx.x) * 4 + k];
        float* d = np_partial + (((long long)n * NSLAB + slab) * P + threadIdx.x) * 4;
        for (int k = 0; k < 4; ++k) d[k] = o[k];
    }
    float v;
    float* gp = p.ws + ((long long)n * NSLAB + slab) * 4;
    v = block_sum_256(kl, red4);   if (threadIdx.x == 0) gp[0] = v;
    v = block_sum_256(ent, red4);  if (threadIdx.x == 0) gp[1] = v;
    v = block_sum_256(patch, red4); if (threadIdx.x == 0) gp[2] = v;
    v = block_sum_256(gmrf, red4); if (threadIdx.x == 0) gp[3] = v;
}

// Two short stages instead of one single-block pass (which walked n * NSLAB * (1 + P) partial records serially per thread: 16 us
// behind a 120 us forward pass).  Stage 1, one block per image: the image's slab partials -> per_np[n][P][8] and the image's eight
// sums (4 pixel sums, 4 sums of squares over its parts), every reduction in a fixed order.  Stage 2, one block: the images.
__global__ __launch_bounds__(256) void prior_finalize_img_kernel(const PriorK p) {
    __shared__ float red4[4];
    const int n = blockIdx.x;
    const float* gpart = p.ws + (long long)n * NSLAB * 4;
    const float* np_partial = p.ws + (long long)p.n * NSLAB * 4 + (long long)n * NSLAB * p.P * 4;
    float* img_part = p.ws + (long long)p.n * NSLAB * 4 + (long long)p.n * NSLAB * p.P * 4 + (long long)n * 8;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < NSLAB; i += 256)
        for (int k = 0; k < 4; ++k) a[k] += gpart[i * 4 + k];
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.view == 0) {
        // wave per part: lanes over the slabs
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int c = wv; c < p.P; c += 4) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = lane; s < NSLAB; s += 64)
                for (int k = 0; k < 4; ++k) o[k] += np_partial[((long long)s * p.P + c) * 4 + k];
            for (int k = 0; k < 4; ++k) o[k] = wave_sum(o[k]);
            if (lane == 0) {
                float* d = p.per_np + ((long long)n * p.P + c) * 8;
                d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3]; d[4] = d[5] = d[6] = d[7] = 0.f;
                sq[0] += o[1] * o[1]; sq[1] += o[0] * o[0]; sq[2] += o[2] * o[2]; sq[3] += o[3] * o[3];
            }
        }
    }
    for (int k = 0; k < 4; ++k) {
        const float v = block_sum_256(a[k], red4);
        if (threadIdx.x == 0) img_part[k] = v;
    }
    for (int k = 0; k < 4; ++k) {
        const float v = block_sum_256(sq[k], red4);
        if (threadIdx.x == 0) img_part[4 + k] = v;
    }
}

__global__ __launch_bounds__(256) void prior_finalize_kernel(const PriorK p) {
    __shared__ float red4[4];
    const float* img_part = p.ws + (long long)p.n * NSLAB * 4 + (long long)p.n * NSLAB * p.P * 4;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < p.n; i += 256)
        for (int k = 0; k < 8; ++k) a[k] += img_part[(long long)i * 8 + k];
    for (int k = 0; k < 8; ++k) {
        const float v = block_sum_256(a[k], red4);
        if (threadIdx.x == 0) p.sums[k] = v;
    }
    if (threadIdx.x == 0) for (int k = 8; k < 16; ++k) p.sums[k] = 0.f;
}

// One block = one tile of tpx (<= 256) consecutive pixels of image blockIdx.y.  LDS: m and (view 0) l_mean with w pixels of
// halo on either side, l / hard / g_hard own pixels only, the per-part constants of the image; the result tiles dl (in the l
// slot) and dl_rec (in the g_hard slot) go back out with 16-byte stores.
template <int PC, int LW>
__global__ __launch_bounds__(256) void prior_bwd_kernel(const PriorK p, int tpx) {
    extern __shared__ __attribute__((aligned(16))) float ts[];
    const int P = PC > 0 ? PC : p.P, PP = tile_pitch(P);
    const int W = LW >= 0 ? (1 << (LW >= 0 ? LW : 0)) : p.w;
    const int hw = p.h * W;
    const int tiles = (hw + tpx - 1) / tpx;
    const int lb = xcd_logical_block();
    const int n = lb / tiles;
    const int t0 = (lb - n * tiles) * tpx;
    const int cnt = min(tpx, hw - t0);
    const int halo = p.view == 0 ? W : 0;
    const int lo = max(0, t0 - halo), hi = min(hw, t0 + cnt + halo);
    const int off = t0 - lo;                       // tile pixel px sits at staged pixel px + off of the halo maps
    float* tm = ts;
    float* tlm = tm + (size_t)(tpx + 2 * halo) * PP;
    float* tl = tlm + (p.view == 0 ? (size_t)(tpx + 2 * halo) * PP : 0);      // l (view 0) -> direct term -> dl
    float* th = tl + (size_t)tpx * PP;                                          // hard (view 0) -> gm
    float* tg = th + (size_t)tpx * PP;                                          // g_hard -> dl_rec
    float* cst = tg + (size_t)tpx * PP;                                         // [P][8]: per-part constants of this image
    const long long img = (long long)n * hw;
    {   // every map's pieces are requested before any of them is waited for: one HBM round trip per block instead of five
        TileReq<4> rm, rlm;
        TileReq<2> rl, rh, rg;
        tile_request(rm, p.m + (img + lo) * P, hi - lo, P);
        if (p.view == 0) {
            tile_request(rlm, p.l_mean + (img + lo) * P, hi - lo, P);
            tile_request(rl, p.l + (img + t0) * P, cnt, P);
            tile_request(rh, p.hard + (img + t0) * P, cnt, P);
        }
        if (p.g_hard) tile_request(rg, p.g_hard + (img + t0) * P, cnt, P);
        tile_commit(rm, p.m + (img + lo) * P, hi - lo, P, PP, tm);
        if (p.view == 0) {
            tile_commit(rlm, p.l_mean + (img + lo) * P, hi - lo, P, PP, tlm);
            tile_commit(rl, p.l + (img + t0) * P, cnt, P, PP, tl);
            tile_commit(rh, p.hard + (img + t0) * P, cnt, P, PP, th);
        }
        if (p.g_hard) tile_commit(rg, p.g_hard + (img + t0) * P, cnt, P, PP, tg);
    }
    for (int c = threadIdx.x; c < P; c += 256) {
        float* k = cst + c * 8;
        const float* np = p.per_np + ((long long)n * P + c) * 8;
        k[0] = p.px ? (float)p.px[((long long)n * P + c) * 2] : 0.f;
        k[1] = p.px ? (float)p.px[((long long)n * P + c) * 2 + 1] : 0.f;
        if (p.view == 0) { k[2] = np[0]; k[3] = np[1]; }
        else {
            const float Z = np[1];
            k[2] = np[0]; k[3] = Z; k[4] = np[3] / Z; k[5] = np[4] / Z; k[6] = np[5] / Z; k[7] = np[6] / Z;
        }
    }
    __syncthreads();
    const long long npix_total = (long long)p.n * hw;
    const float inv_pix = 1.f / (float)npix_total, inv_n = 1.f / (float)p.n;
    const float a8 = p.ms_alpha * 0.125f, a16 = p.ms_alpha * 0.0625f;
    const float sy = p.h > 1 ? 2.f / (float)(p.h - 1) : 0.f, sx = W > 1 ? 2.f / (float)(W - 1) : 0.f;
    // Compute in four short phases (round 4).  The per-pixel form (one thread walks the P parts of its pixel: a serial chain of
    // ~150 dependent instructions per part with half the block idle at 128-pixel tiles) bounded the launch, not its bytes: the
    // element-wise work now runs on ITEMS (pixel, part) spread over all 256 threads (independent items per thread), only the
    // per-pixel reductions (log-sum-exp, the soft-max Jacobian's dot products) stay per pixel.
    float* pst = cst + P * 8;                                   // [tpx][4]: lse, qs, labsum | dot, dot_r
    const int items = cnt * P;
    const int NS = 256 / P;                                     // pixel lanes of the (part, pixel lane) phases
    const int s2 = threadIdx.x / P, c2 = threadIdx.x - s2 * P;
    const int dyy = NS / W, dxx = NS - dyy * W;                 // pixel step NS in (row, column) form
    if (p.view == 0) {
        for (int px = threadIdx.x; px < cnt; px += 256) {       // A: per-pixel log-sum-exp, sum m * log-soft-max, sum of labels
            const float* mrow = tm + (px + off) * PP;
            const float* lrow = tl + px * PP;
            const float* hrow = th + px * PP;
            float mx = -INFINITY;
            for (int c = 0; c < P; ++c) mx = fmaxf(mx, lrow[c]);
            float se = 0.f;
            for (int c = 0; c < P; ++c) se += __expf(lrow[c] - mx);
            const float lse = mx + __logf(se);
            float qs = 0.f, labsum = 0.f;
            for (int c = 0; c < P; ++c) { qs += mrow[c] * (lrow[c] - lse); labsum += hrow[c]; }
            pst[px * 4] = lse; pst[px * 4 + 1] = qs; pst[px * 4 + 2] = labsum;
        }
        __syncthreads();
        // B: d loss / d m (-> hard slot) and the direct term (-> l slot).  Thread = (part c, pixel lane): the part is FIXED per thread
        // (its constants sit in registers, no item -> (pixel, part) division), the pixel coordinates advance incrementally, and the
        // validity of the stencil neighbours is four flags per pixel instead of two range tests per tap.
        if (s2 < NS) {
            const float* kc = cst + c2 * 8;
            const int rcy = (int)kc[0], rcx = (int)kc[1];
            const float kS = kc[2], kR = kc[3];
            int yy = (t0 + s2) / W, xx = (t0 + s2) - yy * W;
            for (int px = s2; px < cnt; px += NS) {
                const int c = c2;
                const int hp = px + off;
                const bool vr = xx + 1 < W, vd = yy + 1 < p.h, vl = xx > 0, vu = yy > 0;
                const float* tmc = tm + hp * PP + c;
                const float mc = tmc[0];
                const float lse = pst[px * 4], qs = pst[px * 4 + 1], labsum = pst[px * 4 + 2];
                const float sl = tl[px * PP + c] - lse;
                const float hv = th[px * PP + c];
                const float gh = p.g_hard ? tg[px * PP + c] : 0.f;
                const float pm = (float)P * mc;
                float gm = p.w_kl * inv_pix * (__logf(pm + 1e-20f) + __fdividef(pm, pm + 1e-20f)) + gh;
                float direct = p.w_entropy * inv_pix * (-mc * (sl - qs));
                if (p.entropy_ce) direct += p.w_entropy * inv_pix * (-(hv - mc * labsum));
                if (p.variant == 0) {      // patch (STE)
                    const bool in_rect = abs(yy - rcy) <= p.half_h && abs(xx - rcx) <= p.half_w;
                    gm += p.w_patch * inv_n * (in_rect ? 0.f : 1.f);
                }
                // area + mumford-shah (cells: own, left neighbour's -- its right value is me --, upper neighbour's -- its down value is me)
                gm += p.w_area * inv_n * 2.f * kS;
                const float m_r = vr ? tmc[PP] : 0.f;
                const float m_d = vd ? tmc[W * PP] : 0.f;
                float dR = 0.f;
                {
                    const float g = a16 * ((mc - m_r) * (mc - m_r) + (mc - m_d) * (mc - m_d));
                    if (g <= p.ms_lambda) dR += a8 * ((mc - m_r) + (mc - m_d));
                }
                if (vl) {
                    const float m_l = tmc[-PP];
                    const float m_ld = vd ? tmc[(W - 1) * PP] : 0.f;
                    const float g = a16 * ((m_l - mc) * (m_l - mc) + (m_l - m_ld) * (m_l - m_ld));
                    if (g <= p.ms_lambda) dR -= a8 * (m_l - mc);
                }
                if (vu) {
                    const float m_u = tmc[-W * PP];
                    const float m_ur = vr ? tmc[(1 - W) * PP] : 0.f;
                    const float g = a16 * ((m_u - m_ur) * (m_u - m_ur) + (m_u - mc) * (m_u - mc));
                    if (g <= p.ms_lambda) dR -= a8 * (m_u - mc);
                }
                gm += p.w_ms * inv_n * 2.f * kR * dR;
                // gmrf on the noise-free logits (same tensor path: l = l_mean + eps)
                const float* tlc = tlm + hp * PP + c;
                const float lm = tlc[0];
                if (p.variant == 1) {
                    // SB_model48c: d/d l_mean of sum min(alpha * g(l_mean), lambda): the stencil above applied to the logits
                    const float l_r = vr ? tlc[PP] : 0.f;
                    const float l_d = vd ? tlc[W * PP] : 0.f;
                    float dL = 0.f;
                    {
                        const float g = a16 * ((lm - l_r) * (lm - l_r) + (lm - l_d) * (lm - l_d));
                        if (g <= p.ms_lambda) dL += a8 * ((lm - l_r) + (lm - l_d));
                    }
                    if (vl) {
                        const float l_l = tlc[-PP];
                        const float l_ld = vd ? tlc[(W - 1) * PP] : 0.f;
                        const float g = a16 * ((l_l - lm) * (l_l - lm) + (l_l - l_ld) * (l_l - l_ld));
                        if (g <= p.ms_lambda) dL -= a8 * (l_l - lm);
                    }
                    if (vu) {
                        const float l_u = tlc[-W * PP];
                        const float l_ur = vr ? tlc[(1 - W) * PP] : 0.f;
                        const float g = a16 * ((l_u - l_ur) * (l_u - l_ur) + (l_u - lm) * (l_u - lm));
                        if (g <= p.ms_lambda) dL -= a8 * (l_u - lm);
                    }
                    direct += p.w_msl * inv_n * dL;
                }
                float gg = 0.f;
                if (vu) gg += lm - tlc[-W * PP];
                if (vd) gg -= tlc[W * PP] - lm;
                if (vl) gg += lm - tlc[-PP];
                if (vr) gg -= tlc[PP] - lm;
                direct += p.w_gmrf * inv_n * gg;
                th[px * PP + c] = gm; tl[px * PP + c] = direct;
                xx += dxx; yy += dyy;
                if (xx >= W) { xx -= W; ++yy; }
            }
        }
    } else {
        if (s2 < NS) {                                          // B (view 1): d loss / d m -> l slot; thread = (part, pixel lane)
            const float* kc = cst + c2 * 8;
            const int rcy = (int)kc[0], rcx = (int)kc[1];
            const float kmax = kc[2], rZ = __fdividef(1.f, kc[3]), muy = kc[4], mux = kc[5], k6 = kc[6], k7 = kc[7];
            // SB_model48c variance (DF:750-776): v_np = S00^2 + S11^2 of c = softmax_hw(gamma*m) (no rectangle, renormalised),
            // S00 = Qy/Z - muy^2, S11 = Qx/Z - mux^2;  SB_model48i: v_np = Q/Z - muy^2 - mux^2 over softmax_hw(gamma*m) * (1-rect)
            const float Qyn = k7, Qxn = k6 - k7;
            const float S00 = Qyn - muy * muy, S11 = Qxn - mux * mux;
            const float T = k6 - 2.f * muy * muy - 2.f * mux * mux;
            const float wv = p.w_var * inv_n * p.gamma * rZ;
            int yy = (t0 + s2) / W, xx = (t0 + s2) - yy * W;
            for (int px = s2; px < cnt; px += NS) {
                const float gy = -1.f + sy * (float)yy, gx = -1.f + sx * (float)xx;
                const float mc = tm[(px + off) * PP + c2];
                const float gh = p.g_hard ? tg[px * PP + c2] : 0.f;
                const float pm = (float)P * mc;
                float gm = p.w_kl * inv_pix * (__logf(pm + 1e-20f) + __fdividef(pm, pm + 1e-20f)) + gh;
                const float sq = __expf(p.gamma * mc - kmax);
                if (p.variant == 1) {
                    const float ay = gy * gy - 2.f * muy * gy - (Qyn - 2.f * muy * muy);
                    const float ax = gx * gx - 2.f * mux * gx - (Qxn - 2.f * mux * mux);
                    gm += wv * sq * 2.f * (S00 * ay + S11 * ax);
                } else {
                    const float kk = (abs(yy - rcy) <= p.half_h && abs(xx - rcx) <= p.half_w) ? 0.f : 1.f;
                    const float a = gy * gy + gx * gx - 2.f * muy * gy - 2.f * mux * gx;
                    gm += wv * sq * (a * kk - T);
                }
                tl[px * PP + c2] = gm;
                xx += dxx; yy += dyy;
                if (xx >= W) { xx -= W; ++yy; }
            }
        }
    }
    __syncthreads();
    const float* gms = p.view == 0 ? th : tl;                   // where phase B left d loss / d m
    for (int px = threadIdx.x; px < cnt; px += 256) {           // C: the soft-max Jacobian's dot products per pixel
        const float* mrow = tm + (px + off) * PP;
        float dot = 0.f, dot_r = 0.f;
        for (int c = 0; c < P; ++c) {
            const float mc = mrow[c];
            dot += mc * gms[px * PP + c];
            if (p.g_hard) dot_r += mc * tg[px * PP + c];
        }
        pst[px * 4] = dot; pst[px * 4 + 1] = dot_r;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {         // D: dl (l slot) and the reconstruction-only dl_rec (g_hard slot)
        const int px = it / P, c = it - px * P;
        const float mc = tm[(px + off) * PP + c];
        const float gh = p.g_hard ? tg[px * PP + c] : 0.f;
        const float gm = gms[px * PP + c];
        const float direct = p.view == 0 ? tl[px * PP + c] : 0.f;
        tl[px * PP + c] = mc * (gm - pst[px * 4]) + direct;
        tg[px * PP + c] = mc * (gh - pst[px * 4 + 1]);
    }
    __syncthreads();
    tile_store_f32(p.dl + (img + t0) * P, cnt, P, PP, tl);
    if (p.dl_rec) tile_store_f32(p.dl_rec + (img + t0) * P, cnt, P, PP, tg);
}

int gp_of(int P) { int g = 2; while (g < P) g *= 2; return g; }

PriorK to_k(const ups_prior_desc* d, float* ws) {
    PriorK k;
    k.n = d->n; k.h = d->h; k.w = d->w; k.P = d->P; k.view = d->view; k.entropy_ce = d->entropy_ce;
    k.half_h = d->half_h; k.half_w = d->half_w; k.gamma = d->gamma; k.ms_alpha = d->ms_alpha; k.ms_lambda = d->ms_lambda;
    k.w_kl = d->w_kl; k.w_entropy = d->w_entropy; k.w_ms = d->w_ms; k.w_area = d->w_area; k.w_patch = d->w_patch;
    k.w_gmrf = d->w_gmrf; k.w_var = d->w_var; k.variant = d->variant; k.w_msl = d->w_ms_logits;
    k.l = d->l; k.l_mean = d->l_mean; k.m = d->m; k.hard = d->hard; k.px = d->px; k.per_np = d->per_np; k.sums = d->sums;
    k.g_hard = d->g_hard; k.dl = d->dl; k.ws = ws; k.dl_rec = d->dl_rec;
    return k;
}

}  // namespace

// Launch KERNEL<PC, LW> for the descriptor's (P, w): specialised instances for the part counts of the shipped / benchmark configs and
// 128- / 256-wide images, the generic <0, -1> instance otherwise.  (Every instance may need more than the 64 KB default of LDS.)
template <typename K, typename... A>
static int prior_launch_one(K kernel, UpsPerDevice& attr, dim3 grid, size_t shm, hipStream_t s, A... args) {
    if (!attr) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return UPS_E_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), shm, s, args...);
    return UPS_OK;
}
#define UPS_PRIOR_CASE(KERNEL, PCV, LWV, ...)                                                            \
    do { static UpsPerDevice at_; const int rc_ = prior_launch_one(KERNEL<PCV, LWV>, at_, __VA_ARGS__); if (rc_ != UPS_OK) return rc_; } while (0)
#define UPS_PRIOR_DISPATCH(KERNEL, grid, shm, ...)                                                       \
    do {                                                                                                 \
        const int lw_ = d->w == 128 ? 7 : (d->w == 256 ? 8 : -1);                                        \
        const int pc_ = (d->P == 10 || d->P == 16 || d->P == 20 || d->P == 25) ? d->P : 0;              \
        if (lw_ < 0 || pc_ == 0) UPS_PRIOR_CASE(KERNEL, 0, -1, grid, shm, s, __VA_ARGS__);               \
        else if (lw_ == 7) {                                                                             \
            if (pc_ == 10) UPS_PRIOR_CASE(KERNEL, 10, 7, grid, shm, s, __VA_ARGS__);                     \
            else if (pc_ == 16) UPS_PRIOR_CASE(KERNEL, 16, 7, grid, shm, s, __VA_ARGS__);                \
            else if (pc_ == 20) UPS_PRIOR_CASE(KERNEL, 20, 7, grid, shm, s, __VA_ARGS__);                \
            else UPS_PRIOR_CASE(KERNEL, 25, 7, grid, shm, s, __VA_ARGS__);                               \
        } else {                                                                                         \
            if (pc_ == 10) UPS_PRIOR_CASE(KERNEL, 10, 8, grid, shm, s, __VA_ARGS__);                     \
            else if (pc_ == 16) UPS_PRIOR_CASE(KERNEL, 16, 8, grid, shm, s, __VA_ARGS__);                \
            else if (pc_ == 20) UPS_PRIOR_CASE(KERNEL, 20, 8, grid, shm, s, __VA_ARGS__);                \
            else UPS_PRIOR_CASE(KERNEL, 25, 8, grid, shm, s, __VA_ARGS__);                               \
        }                                                                                                \
    } while (0)

extern "C" size_t ups_prior_sums_floats(int32_t n, int32_t P) { return 16 + (size_t)n * NSLAB * 4 + (size_t)n * NSLAB * P * 4 + (size_t)n * 8; }

// workspace convention: `sums` points at 16 floats followed by n*NSLAB*4 + n*NSLAB*P*4 + n*8 floats of scratch.
extern "C" int ups_prior_fwd(const ups_prior_desc* d, void* stream) {
    UPS_CHECK_ARG(d && d->m && d->sums && d->P >= 1 && d->P <= 64 && d->n > 0);
    UPS_CHECK_ARG(d->view == 1 || (d->l && d->l_mean && d->hard && (d->px || d->variant == 1) && d->per_np));
    hipStream_t s = (hipStream_t)stream;
    PriorK k = to_k(d, d->sums + 16);
    const int rows = ups_cdiv(d->h, NSLAB);
    const int PP = d->P | 1, NS = 256 / d->P;
    auto lds_fl = [&](int t) {
        return (d->view == 0 ? (size_t)(4 * t + 2 * (d->w + 1)) : (size_t)t) * PP + (size_t)NS * d->P * 4 + 4 + 2 * d->P + (size_t)t;
    };
    int tpx = 256;
    while (tpx > 32 && lds_fl(tpx) * 4 > 48 * 1024) tpx >>= 1;
    const size_t shm = lds_fl(tpx) * sizeof(float);
    UPS_CHECK_ARG(shm <= 160 * 1024);
    UPS_PRIOR_DISPATCH(prior_fwd_kernel, dim3(d->n * NSLAB), shm, k, rows, tpx);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(prior_finalize_img_kernel, dim3(d->n), dim3(256), 0, s, k);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(prior_finalize_kernel, dim3(1), dim3(256), 0, s, k);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_prior_bwd(const ups_prior_desc* d, void* stream) {
    UPS_CHECK_ARG(d && d->m && d->dl && d->per_np && (d->px || d->variant == 1) && d->P >= 1 && d->P <= 64);
    UPS_CHECK_ARG(d->view == 1 || (d->l && d->l_mean && d->hard));
    hipStream_t s = (hipStream_t)stream;
    PriorK k = to_k(d, nullptr);
    const int PP = d->P | 1;
    auto lds_fl = [&](int t) { return ((d->view == 0 ? (size_t)(5 * t + 4 * d->w) : (size_t)(4 * t)) * PP + (size_t)d->P * 8 + (size_t)t * 4); };
    int tpx = 256;
    while (tpx > 32 && lds_fl(tpx) * 4 > 56 * 1024) tpx >>= 1;
    const size_t shm = lds_fl(tpx) * sizeof(float);
    UPS_CHECK_ARG(shm <= 160 * 1024);
    const dim3 grid(ups_cdiv((long long)d->h * d->w, tpx) * d->n);
    UPS_PRIOR_DISPATCH(prior_bwd_kernel, grid, shm, k, tpx);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
