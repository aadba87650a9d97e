// Mask priors of Trainer.make_loss_ops (cub/code/SB_model48i/model.py:652-797) fused into one streaming
// pass per view (forward sums) and one fused backward that emits d(loss)/d(logits) directly.
//
//  view 0 : categorical KL (model.py:21-25,659-665), entropy / cross-entropy (model.py:667-681),
//           Mumford-Shah + area (model.py:744-769, nn.py:1366-1398), patch (model.py:771-783),
//           improper GMRF on the noise-free logits (nn.py:1444-1451)
//  view 1 : categorical KL, variance (model.py:683-719; moments from ups_spatial_moments)
//
// Thread layout: GP = pow2 >= P adjacent lanes own the parts of one pixel (part reductions = shuffles),
// consecutive lane groups own consecutive pixels (coalesced).
//
// sums[16] (written by the forward finalize):
//   0 sum m*log(P*m+1e-20)   1 sum_pix CE/entropy   2 sum hard*(1-rect)   3 sum 0.5*(dy^2+dx^2)
//   4 sum_np R^2   5 sum_np S^2   6 sum_np Rsmooth^2   7 sum_np Rcontour^2
// per_np (view 0) [n][P][8]: 0 S = sum m, 1 R = sum r, 2 Rsmooth, 3 Rcontour
// per_np (view 1) = stats of ups_spatial_moments: 0 max, 1 Z, 2 S0, 3 Sy, 4 Sx, 5 Q
#include "common.h"

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

// ws layout: glob_partial[n][NSLAB][4], then np_partial[n][NSLAB][P][4]
template <int GP>
__global__ __launch_bounds__(256) void prior_fwd_kernel(const PriorK p, int rows_per_slab) {
    constexpr int PL = 256 / GP;
    __shared__ float red_np[PL][GP][4];
    __shared__ float red4[4];
    const int n = blockIdx.x, slab = blockIdx.y;
    const int c = threadIdx.x % GP, pl = threadIdx.x / GP;
    const bool cok = c < p.P;
    const int y0 = slab * rows_per_slab, y1 = min(p.h, y0 + rows_per_slab);
    const long long img = (long long)n * p.h * p.w;
    float kl = 0.f, ent = 0.f, patch = 0.f, gmrf = 0.f;
    float S = 0.f, R = 0.f, Rs = 0.f, Rc = 0.f;
    int cy = 0, cx = 0;
    if (p.view == 0 && cok && p.px) { cy = p.px[((long long)n * p.P + c) * 2]; cx = p.px[((long long)n * p.P + c) * 2 + 1]; }
    const int npix = (y1 - y0) * p.w;
    const int iters = (npix + PL - 1) / PL;
    for (int it = 0; it < iters; ++it) {
        const int q = it * PL + pl;
        const bool pok = q < npix;
        const int yy = y0 + q / p.w, xx = q % p.w;
        const long long pix = img + (long long)yy * p.w + xx;
        const bool ok = pok && cok;
        const float mc = ok ? p.m[pix * p.P + c] : 0.f;
        if (ok) kl += mc * logf((float)p.P * mc + 1e-20f);
        if (p.view == 0) {
            // entropy / CE: needs log-softmax -> group reductions (all lanes participate)
            const float lv = ok ? p.l[pix * p.P + c] : -INFINITY;
            const float mx = gmax<GP>(lv);
            const float se = gsum<GP>(ok ? expf(lv - mx) : 0.f);
            if (ok) {
                const float s = lv - mx - logf(se);
                const float hv = p.hard[pix * p.P + c];
                const float lab = p.entropy_ce ? hv : mc;
                ent += -lab * s;
                const float lm = p.l_mean[pix * p.P + c];
                if (p.variant == 0) {
                    const bool in_rect = abs(yy - cy) <= p.half_h && abs(xx - cx) <= p.half_w;
                    patch += hv * (in_rect ? 0.f : 1.f);
                } else {
                    // SB_model48c: Mumford-Shah on the noise-free logits, min(alpha * g, lambda) summed (patch slot)
                    const float lr = mval(p.l_mean, img, yy, xx + 1, p.h, p.w, p.P, c);
                    const float ld = mval(p.l_mean, img, yy + 1, xx, p.h, p.w, p.P, c);
                    const float gw = 0.25f * (lm - lr), gh = 0.25f * (lm - ld);
                    patch += fminf(p.ms_alpha * (gw * gw + gh * gh), p.ms_lambda);
                }
                if (yy + 1 < p.h) { const float d = p.l_mean[(pix + p.w) * p.P + c] - lm; gmrf += 0.5f * d * d; }
                if (xx + 1 < p.w) { const float d = p.l_mean[(pix + 1) * p.P + c] - lm; gmrf += 0.5f * d * d; }
                const float mr = mval(p.m, img, yy, xx + 1, p.h, p.w, p.P, c);
                const float md = mval(p.m, img, yy + 1, xx, p.h, p.w, p.P, c);
                const float gw = 0.25f * (mc - mr), gh = 0.25f * (mc - md);
                const float g = p.ms_alpha * (gw * gw + gh * gh);
                const float r = fminf(g, p.ms_lambda);
                S += mc; R += r;
                if (g < p.ms_lambda) Rs += r; else Rc += r;
            }
        }
    }
    // per-(n,p) partials
    red_np[pl][c][0] = S; red_np[pl][c][1] = R; red_np[pl][c][2] = Rs; red_np[pl][c][3] = Rc;
    __syncthreads();
    float* np_partial = p.ws + (long long)p.n * NSLAB * 4;
    if (pl == 0 && cok) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < PL; ++q)
            for (int k = 0; k < 4; ++k) o[k] += red_np[q][c][k];
        float* d = np_partial + (((long long)n * NSLAB + slab) * p.P + c) * 4;
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

template <int GP>
__global__ __launch_bounds__(256) void prior_bwd_kernel(const PriorK p) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long pix = gid / GP;
    const int c = (int)(gid % GP);
    const long long hw = (long long)p.h * p.w;
    const long long npix_total = (long long)p.n * hw;
    const bool ok = pix < npix_total && c < p.P;
    const long long pp = ok ? pix : 0;
    const int n = (int)(pp / hw);
    const int q = (int)(pp - (long long)n * hw);
    const int yy = q / p.w, xx = q - yy * p.w;
    const long long img = (long long)n * hw;
    const float inv_pix = 1.f / (float)npix_total, inv_n = 1.f / (float)p.n;

    const float mc = ok ? p.m[pp * p.P + c] : 0.f;
    float gm = 0.f, direct = 0.f, gh = 0.f;
    if (ok) {
        const float pm = (float)p.P * mc;
        gm += p.w_kl * inv_pix * (logf(pm + 1e-20f) + pm / (pm + 1e-20f));
        if (p.g_hard) { gh = p.g_hard[pp * p.P + c]; gm += gh; }
    }
    if (p.view == 0) {
        const float lv = ok ? p.l[pp * p.P + c] : -INFINITY;
        const float mx = gmax<GP>(lv);
        const float se = gsum<GP>(ok ? expf(lv - mx) : 0.f);
        const float s = ok ? lv - mx - logf(se) : 0.f;
        const float hv = ok ? p.hard[pp * p.P + c] : 0.f;
        const float qs = gsum<GP>(mc * s);            // sum_p q_p s_p
        const float labsum = gsum<GP>(hv);
        if (ok) {
            direct += p.w_entropy * inv_pix * (-mc * (s - qs));
            if (p.entropy_ce) direct += p.w_entropy * inv_pix * (-(hv - mc * labsum));
            // patch (STE)
            if (p.variant == 0) {
                const int cy = p.px[((long long)n * p.P + c) * 2], cx = p.px[((long long)n * p.P + c) * 2 + 1];
                const bool in_rect = abs(yy - cy) <= p.half_h && abs(xx - cx) <= p.half_w;
                gm += p.w_patch * inv_n * (in_rect ? 0.f : 1.f);
            }
            // area + mumford-shah
            const float* np = p.per_np + ((long long)n * p.P + c) * 8;
            gm += p.w_area * inv_n * 2.f * np[0];
            const float m_r = mval(p.m, img, yy, xx + 1, p.h, p.w, p.P, c);
            const float m_d = mval(p.m, img, yy + 1, xx, p.h, p.w, p.P, c);
            const float a8 = p.ms_alpha * 0.125f, a16 = p.ms_alpha * 0.0625f;
            float dR = 0.f;
            {   // own cell
                const float g = a16 * ((mc - m_r) * (mc - m_r) + (mc - m_d) * (mc - m_d));
                if (g <= p.ms_lambda) dR += a8 * ((mc - m_r) + (mc - m_d));
            }
            if (xx > 0) {   // left neighbour's cell: its right value is me
                const float m_l = mval(p.m, img, yy, xx - 1, p.h, p.w, p.P, c);
                const float m_ld = mval(p.m, img, yy + 1, xx - 1, p.h, p.w, p.P, c);
                const float g = a16 * ((m_l - mc) * (m_l - mc) + (m_l - m_ld) * (m_l - m_ld));
                if (g <= p.ms_lambda) dR -= a8 * (m_l - mc);
            }
            if (yy > 0) {   // upper neighbour's cell: its down value is me
                const float m_u = mval(p.m, img, yy - 1, xx, p.h, p.w, p.P, c);
                const float m_ur = mval(p.m, img, yy - 1, xx + 1, p.h, p.w, p.P, c);
                const float g = a16 * ((m_u - m_ur) * (m_u - m_ur) + (m_u - mc) * (m_u - mc));
                if (g <= p.ms_lambda) dR -= a8 * (m_u - mc);
            }
            gm += p.w_ms * inv_n * 2.f * np[1] * dR;
            // gmrf on the noise-free logits (same tensor path: l = l_mean + eps)
            const float lm = p.l_mean[pp * p.P + c];
            if (p.variant == 1) {
                // SB_model48c: d/d l_mean of sum min(alpha * g(l_mean), lambda): the stencil above applied to the logits
                const float l_r = mval(p.l_mean, img, yy, xx + 1, p.h, p.w, p.P, c);
                const float l_d = mval(p.l_mean, img, yy + 1, xx, p.h, p.w, p.P, c);
                float dL = 0.f;
                {
                    const float g = a16 * ((lm - l_r) * (lm - l_r) + (lm - l_d) * (lm - l_d));
                    if (g <= p.ms_lambda) dL += a8 * ((lm - l_r) + (lm - l_d));
                }
                if (xx > 0) {
                    const float l_l = mval(p.l_mean, img, yy, xx - 1, p.h, p.w, p.P, c);
                    const float l_ld = mval(p.l_mean, img, yy + 1, xx - 1, p.h, p.w, p.P, c);
                    const float g = a16 * ((l_l - lm) * (l_l - lm) + (l_l - l_ld) * (l_l - l_ld));
                    if (g <= p.ms_lambda) dL -= a8 * (l_l - lm);
                }
                if (yy > 0) {
                    const float l_u = mval(p.l_mean, img, yy - 1, xx, p.h, p.w, p.P, c);
                    const float l_ur = mval(p.l_mean, img, yy - 1, xx + 1, p.h, p.w, p.P, c);
                    const float g = a16 * ((l_u - l_ur) * (l_u - l_ur) + (l_u - lm) * (l_u - lm));
                    if (g <= p.ms_lambda) dL -= a8 * (l_u - lm);
                }
                direct += p.w_msl * inv_n * dL;
            }
            float gg = 0.f;
            if (yy > 0) gg += lm - p.l_mean[(pp - p.w) * p.P + c];
            if (yy + 1 < p.h) gg -= p.l_mean[(pp + p.w) * p.P + c] - lm;
            if (xx > 0) gg += lm - p.l_mean[(pp - 1) * p.P + c];
            if (xx + 1 < p.w) gg -= p.l_mean[(pp + 1) * p.P + c] - lm;
            direct += p.w_gmrf * inv_n * gg;
        }
    } else if (ok && p.variant == 1) {
        // SB_model48c variance (DF:750-776): v_np = S00^2 + S11^2 of c = softmax_hw(gamma*m) (no rectangle, renormalised),
        // S00 = Qy/Z - muy^2, S11 = Qx/Z - mux^2
        const float* st = p.per_np + ((long long)n * p.P + c) * 8;
        const float Z = st[1], muy = st[3] / Z, mux = st[4] / Z, Qyn = st[6] / Z, Qxn = (st[5] - st[6]) / Z;
        const float S00 = Qyn - muy * muy, S11 = Qxn - mux * mux;
        const float sy = p.h > 1 ? 2.f / (float)(p.h - 1) : 0.f, sx = p.w > 1 ? 2.f / (float)(p.w - 1) : 0.f;
        const float gy = -1.f + sy * (float)yy, gx = -1.f + sx * (float)xx;
        const float ay = gy * gy - 2.f * muy * gy - (Qyn - 2.f * muy * muy);
        const float ax = gx * gx - 2.f * mux * gx - (Qxn - 2.f * mux * mux);
        const float sq = expf(p.gamma * mc - st[0]) / Z;
        gm += p.w_var * inv_n * p.gamma * sq * 2.f * (S00 * ay + S11 * ax);
    } else if (ok) {
        // variance: v_np = Q/Z - muy^2 - mux^2 over c = softmax_hw(gamma*m) * (1-rect)
        const float* st = p.per_np + ((long long)n * p.P + c) * 8;
        const float Z = st[1], muy = st[3] / Z, mux = st[4] / Z, Qn = st[5] / Z;
        const float T = Qn - 2.f * muy * muy - 2.f * mux * mux;
        const float sy = p.h > 1 ? 2.f / (float)(p.h - 1) : 0.f, sx = p.w > 1 ? 2.f / (float)(p.w - 1) : 0.f;
        const float gy = -1.f + sy * (float)yy, gx = -1.f + sx * (float)xx;
        const int cy = p.px[((long long)n * p.P + c) * 2], cx = p.px[((long long)n * p.P + c) * 2 + 1];
        const float k = (abs(yy - cy) <= p.half_h && abs(xx - cx) <= p.half_w) ? 0.f : 1.f;
        const float a = gy * gy + gx * gx - 2.f * muy * gy - 2.f * mux * gx;
        const float sq = expf(p.gamma * mc - st[0]) / Z;
        gm += p.w_var * inv_n * p.gamma * sq * (a * k - T);
    }
    const float dot = gsum<GP>(mc * gm);
    if (ok) p.dl[pp * p.P + c] = mc * (gm - dot) + direct;
    if (p.dl_rec) {     // the same launch also emits the gradient of the reconstruction loss alone (all prior weights zero)
        const float dot_r = gsum<GP>(mc * gh);
        if (ok) p.dl_rec[pp * p.P + c] = mc * (gh - dot_r);
    }
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
    const int gp = gp_of(d->P);
#define UPS_PF(G) hipLaunchKernelGGL(prior_fwd_kernel<G>, dim3(d->n, NSLAB), dim3(256), 0, s, k, rows)
    switch (gp) {
        case 2: UPS_PF(2); break; case 4: UPS_PF(4); break; case 8: UPS_PF(8); break;
        case 16: UPS_PF(16); break; case 32: UPS_PF(32); break; default: UPS_PF(64); break;
    }
#undef UPS_PF
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
    const int gp = gp_of(d->P);
    const long long threads = (long long)d->n * d->h * d->w * gp;
    const int grid = ups_cdiv(threads, 256);
#define UPS_PB(G) hipLaunchKernelGGL(prior_bwd_kernel<G>, dim3(grid), dim3(256), 0, s, k)
    switch (gp) {
        case 2: UPS_PB(2); break; case 4: UPS_PB(4); break; case 8: UPS_PB(8); break;
        case 16: UPS_PB(16); break; case 32: UPS_PB(32); break; default: UPS_PB(64); break;
    }
#undef UPS_PB
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
