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

constexpr int NSLAB = 32;     // row slabs per image in the forward pass (n x NSLAB blocks: 2 048 at B = 64)

struct PriorK {
    int n, h, w, P, view, entropy_ce, half_h, half_w, variant;
    float gamma, ms_alpha, ms_lambda, w_kl, w_entropy, w_ms, w_area, w_patch, w_gmrf, w_var, w_msl;
    const float* l; const float* l_mean; const float* m; const float* hard; const int* px;
    float* per_np; float* sums; const float* g_hard; float* dl; float* ws; float* dl_rec;
};

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
__global__ __launch_bounds__(256) void prior_fwd_kernel(const PriorK p, int rows_per_slab, int tpx) {
    extern __shared__ __attribute__((aligned(16))) float ts[];
    const int P = p.P, PP = tile_pitch(P);
    const int halo = p.view == 0 ? p.w + 1 : 0;
    float* tm = ts;
    float* tlm = tm + (size_t)(tpx + halo) * PP;
    float* tl = tlm + (p.view == 0 ? (size_t)(tpx + halo) * PP : 0);
    float* th = tl + (p.view == 0 ? (size_t)tpx * PP : 0);
    float* scratch = th + (p.view == 0 ? (size_t)tpx * PP : 0);      // red_np[NS][P][4], red4[4]
    const int NS = 256 / P;                                           // sub-lanes per part in phase 2
    float* red4 = scratch + (size_t)NS * P * 4;
    int* cpx = (int*)(red4 + 4);                                      // rectangle centres of this image [P][2]
    const int n = blockIdx.x, slab = blockIdx.y;
    if (p.view == 0 && p.px)
        for (int i = threadIdx.x; i < 2 * P; i += 256) cpx[i] = p.px[(long long)n * P * 2 + i];
    const int y0 = slab * rows_per_slab, y1 = max(y0, min(p.h, y0 + rows_per_slab));
    const int hw = p.h * p.w;
    const long long img = (long long)n * hw;
    float kl = 0.f, ent = 0.f, patch = 0.f, gmrf = 0.f;
    float S = 0.f, R = 0.f, Rs = 0.f, Rc = 0.f;
    const int c2 = threadIdx.x % P, s2 = threadIdx.x / P;            // phase-2 role
    const int q0 = y0 * p.w, q1 = y1 * p.w;
    for (int t0 = q0; t0 < q1; t0 += tpx) {
        const int cnt = min(tpx, q1 - t0);
        const int cnt_h = min(cnt + halo, hw - t0);
        __syncthreads();
        tile_load_f32(p.m + (img + t0) * P, cnt_h, P, PP, tm);
        if (p.view == 0) {
            tile_load_f32(p.l_mean + (img + t0) * P, cnt_h, P, PP, tlm);
            tile_load_f32(p.l + (img + t0) * P, cnt, P, PP, tl);
            tile_load_f32(p.hard + (img + t0) * P, cnt, P, PP, th);
        }
        __syncthreads();
        for (int px = threadIdx.x; px < cnt; px += 256) {
            const int q = t0 + px;
            const int yy = q / p.w, xx = q - yy * p.w;
            const float* mrow = tm + px * PP;
            for (int c = 0; c < P; ++c) { const float mc = mrow[c]; kl += mc * logf((float)P * mc + 1e-20f); }
            if (p.view == 0) {
                const float* lrow = tl + px * PP;
                float* hrow = th + px * PP;
                float mx = -INFINITY;
                for (int c = 0; c < P; ++c) mx = fmaxf(mx, lrow[c]);
                float se = 0.f;
                for (int c = 0; c < P; ++c) se += expf(lrow[c] - mx);
                const float lse = mx + logf(se);
                for (int c = 0; c < P; ++c) {
                    const float mc = mrow[c];
                    const float sl = lrow[c] - lse;
                    const float hv = hrow[c];
                    ent += -(p.entropy_ce ? hv : mc) * sl;
                    const float lm = tlm[px * PP + c];
                    if (p.variant == 0) {
                        const bool in_rect = abs(yy - cpx[2 * c]) <= p.half_h && abs(xx - cpx[2 * c + 1]) <= p.half_w;
                        patch += hv * (in_rect ? 0.f : 1.f);
                    } else {
                        // SB_model48c: Mumford-Shah on the noise-free logits, min(alpha * g, lambda) summed (patch slot)
                        const float lr = tval(tlm, PP, px, 1, c, yy, xx + 1, p.h, p.w);
                        const float ld = tval(tlm, PP, px, p.w, c, yy + 1, xx, p.h, p.w);
                        const float gw = 0.25f * (lm - lr), gh = 0.25f * (lm - ld);
                        patch += fminf(p.ms_alpha * (gw * gw + gh * gh), p.ms_lambda);
                    }
                    if (yy + 1 < p.h) { const float d = tlm[(px + p.w) * PP + c] - lm; gmrf += 0.5f * d * d; }
                    if (xx + 1 < p.w) { const float d = tlm[(px + 1) * PP + c] - lm; gmrf += 0.5f * d * d; }
                    const float mr = tval(tm, PP, px, 1, c, yy, xx + 1, p.h, p.w);
                    const float md = tval(tm, PP, px, p.w, c, yy + 1, xx, p.h, p.w);
                    const float gw = 0.25f * (mc - mr), gh = 0.25f * (mc - md);
                    const float g = p.ms_alpha * (gw * gw + gh * gh);
                    const float r = fminf(g, p.ms_lambda);
                    hrow[c] = (g < p.ms_lambda) ? r : -r;            // sign = contour branch (r = lambda > 0 there)
                }
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

__global__ __launch_bounds__(256) void prior_finalize_kernel(const PriorK p) {
    __shared__ float red4[4];
    const float* gpart = p.ws;
    const float* np_partial = p.ws + (long long)p.n * NSLAB * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < p.n * NSLAB; i += 256)
        for (int k = 0; k < 4; ++k) a[k] += gpart[(long long)i * 4 + k];
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.view == 0) {
        for (int i = threadIdx.x; i < p.n * p.P; i += 256) {
            const int n = i / p.P, c = i - n * p.P;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < NSLAB; ++s)
                for (int k = 0; k < 4; ++k) o[k] += np_partial[(((long long)n * NSLAB + s) * p.P + c) * 4 + k];
            float* d = p.per_np + (long long)i * 8;
            d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3]; d[4] = d[5] = d[6] = d[7] = 0.f;
            sq[0] += o[1] * o[1]; sq[1] += o[0] * o[0]; sq[2] += o[2] * o[2]; sq[3] += o[3] * o[3];
        }
    }
    for (int k = 0; k < 4; ++k) {
        const float v = block_sum_256(a[k], red4);
        if (threadIdx.x == 0) p.sums[k] = v;
    }
    for (int k = 0; k < 4; ++k) {
        const float v = block_sum_256(sq[k], red4);
        if (threadIdx.x == 0) p.sums[4 + k] = v;
    }
    if (threadIdx.x == 0) for (int k = 8; k < 16; ++k) p.sums[k] = 0.f;
}

// One block = one tile of tpx (<= 256) consecutive pixels of image blockIdx.y.  LDS: m and (view 0) l_mean with w pixels of
// halo on either side, l / hard / g_hard own pixels only, the per-part constants of the image; the result tiles dl (in the l
// slot) and dl_rec (in the g_hard slot) go back out with 16-byte stores.
__global__ __launch_bounds__(256) void prior_bwd_kernel(const PriorK p, int tpx) {
    extern __shared__ __attribute__((aligned(16))) float ts[];
    const int P = p.P, PP = tile_pitch(P);
    const int n = blockIdx.y;
    const int hw = p.h * p.w;
    const int t0 = blockIdx.x * tpx;
    const int cnt = min(tpx, hw - t0);
    const int halo = p.view == 0 ? p.w : 0;
    const int lo = max(0, t0 - halo), hi = min(hw, t0 + cnt + halo);
    const int off = t0 - lo;                       // tile pixel px sits at staged pixel px + off of the halo maps
    float* tm = ts;
    float* tlm = tm + (size_t)(tpx + 2 * halo) * PP;
    float* tl = tlm + (p.view == 0 ? (size_t)(tpx + 2 * halo) * PP : 0);      // l (view 0) -> direct term -> dl
    float* th = tl + (size_t)tpx * PP;                                          // hard (view 0) -> gm
    float* tg = th + (size_t)tpx * PP;                                          // g_hard -> dl_rec
    float* cst = tg + (size_t)tpx * PP;                                         // [P][8]: per-part constants of this image
    const long long img = (long long)n * hw;
    tile_load_f32(p.m + (img + lo) * P, hi - lo, P, PP, tm);
    if (p.view == 0) {
        tile_load_f32(p.l_mean + (img + lo) * P, hi - lo, P, PP, tlm);
        tile_load_f32(p.l + (img + t0) * P, cnt, P, PP, tl);
        tile_load_f32(p.hard + (img + t0) * P, cnt, P, PP, th);
    }
    if (p.g_hard) tile_load_f32(p.g_hard + (img + t0) * P, cnt, P, PP, tg);
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
    const float sy = p.h > 1 ? 2.f / (float)(p.h - 1) : 0.f, sx = p.w > 1 ? 2.f / (float)(p.w - 1) : 0.f;
    for (int px = threadIdx.x; px < cnt; px += 256) {
        const int q = t0 + px;
        const int yy = q / p.w, xx = q - yy * p.w;
        const int hp = px + off;
        const float* mrow = tm + hp * PP;
        float* lrow = tl + px * PP;
        float* hrow = th + px * PP;
        float* grow = tg + px * PP;
        float dot = 0.f, dot_r = 0.f;
        if (p.view == 0) {
            float mx = -INFINITY;
            for (int c = 0; c < P; ++c) mx = fmaxf(mx, lrow[c]);
            float se = 0.f;
            for (int c = 0; c < P; ++c) se += expf(lrow[c] - mx);
            const float lse = mx + logf(se);
            float qs = 0.f, labsum = 0.f;
            for (int c = 0; c < P; ++c) { qs += mrow[c] * (lrow[c] - lse); labsum += hrow[c]; }
            for (int c = 0; c < P; ++c) {
                const float mc = mrow[c];
                const float sl = lrow[c] - lse;
                const float hv = hrow[c];
                const float gh = p.g_hard ? grow[c] : 0.f;
                const float pm = (float)P * mc;
                float gm = p.w_kl * inv_pix * (logf(pm + 1e-20f) + pm / (pm + 1e-20f)) + gh;
                float direct = p.w_entropy * inv_pix * (-mc * (sl - qs));
                if (p.entropy_ce) direct += p.w_entropy * inv_pix * (-(hv - mc * labsum));
                const float* k = cst + c * 8;
                if (p.variant == 0) {      // patch (STE)
                    const bool in_rect = abs(yy - (int)k[0]) <= p.half_h && abs(xx - (int)k[1]) <= p.half_w;
                    gm += p.w_patch * inv_n * (in_rect ? 0.f : 1.f);
                }
                // area + mumford-shah
                gm += p.w_area * inv_n * 2.f * k[2];
                const float m_r = tval(tm, PP, hp, 1, c, yy, xx + 1, p.h, p.w);
                const float m_d = tval(tm, PP, hp, p.w, c, yy + 1, xx, p.h, p.w);
                float dR = 0.f;
                {   // own cell
                    const float g = a16 * ((mc - m_r) * (mc - m_r) + (mc - m_d) * (mc - m_d));
                    if (g <= p.ms_lambda) dR += a8 * ((mc - m_r) + (mc - m_d));
                }
                if (xx > 0) {   // left neighbour's cell: its right value is me
                    const float m_l = tval(tm, PP, hp, -1, c, yy, xx - 1, p.h, p.w);
                    const float m_ld = tval(tm, PP, hp, p.w - 1, c, yy + 1, xx - 1, p.h, p.w);
                    const float g = a16 * ((m_l - mc) * (m_l - mc) + (m_l - m_ld) * (m_l - m_ld));
                    if (g <= p.ms_lambda) dR -= a8 * (m_l - mc);
                }
                if (yy > 0) {   // upper neighbour's cell: its down value is me
                    const float m_u = tval(tm, PP, hp, -p.w, c, yy - 1, xx, p.h, p.w);
                    const float m_ur = tval(tm, PP, hp, -p.w + 1, c, yy - 1, xx + 1, p.h, p.w);
                    const float g = a16 * ((m_u - m_ur) * (m_u - m_ur) + (m_u - mc) * (m_u - mc));
                    if (g <= p.ms_lambda) dR -= a8 * (m_u - mc);
                }
                gm += p.w_ms * inv_n * 2.f * k[3] * dR;
                // gmrf on the noise-free logits (same tensor path: l = l_mean + eps)
                const float lm = tlm[hp * PP + c];
                if (p.variant == 1) {
                    // SB_model48c: d/d l_mean of sum min(alpha * g(l_mean), lambda): the stencil above applied to the logits
                    const float l_r = tval(tlm, PP, hp, 1, c, yy, xx + 1, p.h, p.w);
                    const float l_d = tval(tlm, PP, hp, p.w, c, yy + 1, xx, p.h, p.w);
                    float dL = 0.f;
                    {
                        const float g = a16 * ((lm - l_r) * (lm - l_r) + (lm - l_d) * (lm - l_d));
                        if (g <= p.ms_lambda) dL += a8 * ((lm - l_r) + (lm - l_d));
                    }
                    if (xx > 0) {
                        const float l_l = tval(tlm, PP, hp, -1, c, yy, xx - 1, p.h, p.w);
                        const float l_ld = tval(tlm, PP, hp, p.w - 1, c, yy + 1, xx - 1, p.h, p.w);
                        const float g = a16 * ((l_l - lm) * (l_l - lm) + (l_l - l_ld) * (l_l - l_ld));
                        if (g <= p.ms_lambda) dL -= a8 * (l_l - lm);
                    }
                    if (yy > 0) {
                        const float l_u = tval(tlm, PP, hp, -p.w, c, yy - 1, xx, p.h, p.w);
                        const float l_ur = tval(tlm, PP, hp, -p.w + 1, c, yy - 1, xx + 1, p.h, p.w);
                        const float g = a16 * ((l_u - l_ur) * (l_u - l_ur) + (l_u - lm) * (l_u - lm));
                        if (g <= p.ms_lambda) dL -= a8 * (l_u - lm);
                    }
                    direct += p.w_msl * inv_n * dL;
                }
                float gg = 0.f;
                if (yy > 0) gg += lm - tlm[(hp - p.w) * PP + c];
                if (yy + 1 < p.h) gg -= tlm[(hp + p.w) * PP + c] - lm;
                if (xx > 0) gg += lm - tlm[(hp - 1) * PP + c];
                if (xx + 1 < p.w) gg -= tlm[(hp + 1) * PP + c] - lm;
                direct += p.w_gmrf * inv_n * gg;
                dot += mc * gm; dot_r += mc * gh;
                hrow[c] = gm; lrow[c] = direct;
            }
            for (int c = 0; c < P; ++c) {
                const float mc = mrow[c];
                const float gh = p.g_hard ? grow[c] : 0.f;
                lrow[c] = mc * (hrow[c] - dot) + lrow[c];
                // the same launch also emits the gradient of the reconstruction loss alone (all prior weights zero)
                grow[c] = mc * (gh - dot_r);
            }
        } else {
            const float gy = -1.f + sy * (float)yy, gx = -1.f + sx * (float)xx;
            for (int c = 0; c < P; ++c) {
                const float mc = mrow[c];
                const float gh = p.g_hard ? grow[c] : 0.f;
                const float pm = (float)P * mc;
                float gm = p.w_kl * inv_pix * (logf(pm + 1e-20f) + pm / (pm + 1e-20f)) + gh;
                const float* k = cst + c * 8;
                const float Z = k[3], muy = k[4], mux = k[5];
                const float sq = expf(p.gamma * mc - k[2]) / Z;
                if (p.variant == 1) {
                    // SB_model48c variance (DF:750-776): v_np = S00^2 + S11^2 of c = softmax_hw(gamma*m) (no rectangle,
                    // renormalised), S00 = Qy/Z - muy^2, S11 = Qx/Z - mux^2
                    const float Qyn = k[7], Qxn = k[6] - k[7];
                    const float S00 = Qyn - muy * muy, S11 = Qxn - mux * mux;
                    const float ay = gy * gy - 2.f * muy * gy - (Qyn - 2.f * muy * muy);
                    const float ax = gx * gx - 2.f * mux * gx - (Qxn - 2.f * mux * mux);
                    gm += p.w_var * inv_n * p.gamma * sq * 2.f * (S00 * ay + S11 * ax);
                } else {
                    // variance: v_np = Q/Z - muy^2 - mux^2 over c = softmax_hw(gamma*m) * (1-rect)
                    const float Qn = k[6];
                    const float T = Qn - 2.f * muy * muy - 2.f * mux * mux;
                    const float kk = (abs(yy - (int)k[0]) <= p.half_h && abs(xx - (int)k[1]) <= p.half_w) ? 0.f : 1.f;
                    const float a = gy * gy + gx * gx - 2.f * muy * gy - 2.f * mux * gx;
                    gm += p.w_var * inv_n * p.gamma * sq * (a * kk - T);
                }
                dot += mc * gm; dot_r += mc * gh;
                lrow[c] = gm;
            }
            for (int c = 0; c < P; ++c) {
                const float mc = mrow[c];
                const float gh = p.g_hard ? grow[c] : 0.f;
                lrow[c] = mc * (lrow[c] - dot);
                grow[c] = mc * (gh - dot_r);
            }
        }
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

extern "C" size_t ups_prior_sums_floats(int32_t n, int32_t P) { return 16 + (size_t)n * NSLAB * 4 + (size_t)n * NSLAB * P * 4; }

// workspace convention: `sums` points at 16 floats followed by n*NSLAB*4 + n*NSLAB*P*4 floats of scratch.
extern "C" int ups_prior_fwd(const ups_prior_desc* d, void* stream) {
    UPS_CHECK_ARG(d && d->m && d->sums && d->P >= 1 && d->P <= 64 && d->n > 0);
    UPS_CHECK_ARG(d->view == 1 || (d->l && d->l_mean && d->hard && (d->px || d->variant == 1) && d->per_np));
    hipStream_t s = (hipStream_t)stream;
    PriorK k = to_k(d, d->sums + 16);
    const int rows = ups_cdiv(d->h, NSLAB);
    const int PP = d->P | 1, NS = 256 / d->P;
    auto lds_fl = [&](int t) {
        return (d->view == 0 ? (size_t)(4 * t + 2 * (d->w + 1)) : (size_t)t) * PP + (size_t)NS * d->P * 4 + 4 + 2 * d->P;
    };
    int tpx = 256;
    while (tpx > 32 && lds_fl(tpx) * 4 > 48 * 1024) tpx >>= 1;
    const size_t shm = lds_fl(tpx) * sizeof(float);
    UPS_CHECK_ARG(shm <= 160 * 1024);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)prior_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return UPS_E_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(prior_fwd_kernel, dim3(d->n, NSLAB), dim3(256), shm, s, k, rows, tpx);
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
    auto lds_fl = [&](int t) { return ((d->view == 0 ? (size_t)(5 * t + 4 * d->w) : (size_t)(4 * t)) * PP + (size_t)d->P * 8); };
    int tpx = 256;
    while (tpx > 32 && lds_fl(tpx) * 4 > 56 * 1024) tpx >>= 1;
    const size_t shm = lds_fl(tpx) * sizeof(float);
    UPS_CHECK_ARG(shm <= 160 * 1024);
    static bool attr = false;
    if (!attr) {      // wide images (large w * P) can need more than the 64 KB default
        if (hipFuncSetAttribute((const void*)prior_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return UPS_E_LAUNCH;
        attr = true;
    }
    const dim3 grid(ups_cdiv((long long)d->h * d->w, tpx), d->n);
    hipLaunchKernelGGL(prior_bwd_kernel, grid, dim3(256), shm, s, k, tpx);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
