// Full-covariance latent (cub/code/nn.py:1134-1208 + the spiral fill of cub/code/util.py:878-995),
// TF-style Adam (tf.train.AdamOptimizer), the Gaussian renderers of nn.py:1639-1702
// and nn.py:1976-2021, and the error plumbing of the C ABI.
#include <stdarg.h>

#include "common.h"

// ------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
void ups_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ups_last_error(void) { return g_err; }
extern "C" int ups_abi_version(void) { return UPS_ABI_VERSION; }
extern "C" void ups_struct_sizes(int64_t out[4]) {
    out[0] = (int64_t)sizeof(ups_conv_desc); out[1] = (int64_t)sizeof(ups_wgrad_desc);
    out[2] = (int64_t)sizeof(ups_prior_desc); out[3] = (int64_t)sizeof(ups_prep_item);
}

namespace {

constexpr int MAXS = 12;

// index into the triangular parameter vector x (length m = n(n+1)/2) of element (i, j<=i) of
// fill_triangular(x) (util.py:981-993): row-major position q of concat(x[n:], reverse(x))
__device__ inline int tri_index(int i, int j, int n, int m) {
    const int q = i * n + j;
    return q < m - n ? n + q : m - 1 - (q - (m - n));
}

struct Levels { float v[MAXS]; };

// one wave per matrix row; lanes stride over the columns (contiguous parameter reads)
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ params, const float* __restrict__ eps,
                                                         Levels lv, int S, int B, int n, float* __restrict__ samples,
                                                         float* __restrict__ kl_rows) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int m = n * (n + 1) / 2, np = n + m;
    const float* pb = params + (long long)b * np;
    for (int i = blockIdx.y * 4 + wid; i < n; i += gridDim.y * 4) {
        const float rs = 1.f / sqrtf((float)(i + 1));
        float dot[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) dot[s] = 0.f;
        float sumsq = 0.f;
        for (int j = lane; j <= i; j += 64) {
            const float raw = pb[n + tri_index(i, j, n, m)];
            const float L = (j == i) ? expf(raw) : raw * rs;
            sumsq += L * L;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) dot[s] += L * lv.v[s] * eps[((long long)s * B + b) * n + j];
        }
        sumsq = wave_sum(sumsq);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < S) dot[s] = wave_sum(dot[s]);
        if (lane == 0) {
            const float mean = pb[i];
            const float raw_ii = pb[n + tri_index(i, i, n, m)];
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) samples[((long long)s * B + b) * n + i] = mean + dot[s];
            if (kl_rows) kl_rows[(long long)b * n + i] = 0.5f * (sumsq - 1.f + mean * mean - 2.f * raw_ii);
        }
    }
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ params, const float* __restrict__ eps,
                                                         Levels lv, const float* __restrict__ gs,
                                                         const float* __restrict__ gk_dev, float gk_scale, int S, int B,
                                                         int n, float* __restrict__ gp) {
    const float gk = gk_scale * (gk_dev ? gk_dev[0] : 1.f);
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int m = n * (n + 1) / 2, np = n + m;
    const float* pb = params + (long long)b * np;
    float* gb = gp + (long long)b * np;
    for (int i = blockIdx.y * 4 + wid; i < n; i += gridDim.y * 4) {
        const float rs = 1.f / sqrtf((float)(i + 1));
        float gi[MAXS];
        float gsum = 0.f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            gi[s] = (s < S) ? gs[((long long)s * B + b) * n + i] : 0.f;
            gsum += gi[s];
            gi[s] *= lv.v[s < S ? s : 0];
        }
        for (int j = lane; j <= i; j += 64) {
            const int idx = n + tri_index(i, j, n, m);
            const float raw = pb[idx];
            const float L = (j == i) ? expf(raw) : raw * rs;
            float dL = gk * L;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) dL += gi[s] * eps[((long long)s * B + b) * n + j];
            gb[idx] = (j == i) ? dL * L - gk : dL * rs;
        }
        if (lane == 0) gb[i] = gsum + gk * pb[i];
    }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long count, float lr_t, const float* __restrict__ lr_t_dev, float b1, float b2, float eps,
                            float gscale) {
    if (lr_t_dev) lr_t = *lr_t_dev;       // device scalar: lets a captured HIP graph be replayed with the step's learning rate
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const float gg = g[i] * gscale;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        p[i] = p[i] - lr_t * mm / (sqrtf(vv) + eps);
    }
}

__global__ void gauss_hm_kernel(const float* __restrict__ pts, const float* __restrict__ sd, float* __restrict__ out, int B,
                                int h, int w, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * h * w * K) return;
    const int k = (int)(idx % K);
    long long t = idx / K;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    const float* pp = pts + ((long long)b * K + k) * 2;
    const float* ss = sd + ((long long)b * K + k) * 2;
    const float dx = (float)x - pp[0], dy = (float)y - pp[1];
    out[idx] = expf(-(dx * dx) / (2.f * ss[0] * ss[0]) - (dy * dy) / (2.f * ss[1] * ss[1]));
}

__global__ void gauss_hm3_kernel(const float* __restrict__ mu, const float* __restrict__ L, float* __restrict__ out, int B,
                                 int h, int w, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * h * w * K) return;
    const int k = (int)(idx % K);
    long long t = idx / K;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    const float* m2 = mu + ((long long)b * K + k) * 2;
    const float* l4 = L + ((long long)b * K + k) * 4;
    const float gy = h > 1 ? -1.f + 2.f * (float)y / (float)(h - 1) : -1.f;
    const float gx = w > 1 ? -1.f + 2.f * (float)x / (float)(w - 1) : -1.f;
    const float d0 = gy - m2[0], d1 = gx - m2[1];
    const float z0 = d0 / l4[0];
    const float z1 = (d1 - l4[2] * z0) / l4[3];
    out[idx] = expf(-0.5f * (z0 * z0 + z1 * z1)) / (6.283185307179586f * fabsf(l4[0] * l4[3]));
}


// ------------------------------------------------------------------ standard-normal noise (tf.random_normal, nn.py:1187,1431)
// Philox4x32-10 (Salmon et al. 2011) keyed by `seed`, counter = offset + index of the 4-value group; the four 32-bit words of a
// group -> two Box-Muller pairs.  A pure function of (seed, offset, element index): the same stream under any launch geometry
// and under HIP-graph replay (the trainer advances `offset` by ceil(n / 4) per call).
__device__ inline void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ void randn_kernel(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long offset) {
    const long long groups = (n + 3) >> 2;
    for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += (long long)gridDim.x * blockDim.x) {
        const unsigned long long ctr = offset + (unsigned long long)gidx;
        unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1): 24 bits, never 0
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
        }
        const long long e = gidx << 2;
        if (e + 3 < n && ((((unsigned long long)out) & 15ull) == 0)) *(float4*)(out + e) = make_float4(z[0], z[1], z[2], z[3]);
        else for (int k = 0; k < 4 && e + k < n; ++k) out[e + k] = z[k];
    }
}

}  // namespace

extern "C" int ups_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    UPS_CHECK_ARG(out && n > 0);
    long long grid = ((n + 3) / 4 + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, out, (long long)n, (unsigned long long)seed,
                       (unsigned long long)offset);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_latent_fwd(const float* params, const float* eps, const float* level, int32_t S, int32_t B, int32_t dim,
                              float* samples, float* kl_rows, void* stream) {
    UPS_CHECK_ARG(params && eps && level && samples && S >= 1 && S <= MAXS && B > 0 && dim > 0);
    Levels lv;
    for (int s = 0; s < MAXS; ++s) lv.v[s] = s < S ? level[s] : 0.f;   // `level` is a HOST array
    hipLaunchKernelGGL(latent_fwd_kernel, dim3(B, ups_cdiv(dim, 32)), dim3(256), 0, (hipStream_t)stream, params, eps, lv, S,
                       B, dim, samples, kl_rows);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_latent_bwd(const float* params, const float* eps, const float* level, const float* g_samples,
                              const float* g_kl_dev, float g_kl_scale, int32_t S, int32_t B, int32_t dim, float* g_params,
                              void* stream) {
    UPS_CHECK_ARG(params && eps && level && g_samples && g_params && S >= 1 && S <= MAXS);
    Levels lv;
    for (int s = 0; s < MAXS; ++s) lv.v[s] = s < S ? level[s] : 0.f;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3(B, ups_cdiv(dim, 32)), dim3(256), 0, (hipStream_t)stream, params, eps, lv,
                       g_samples, g_kl_dev, g_kl_scale, S, B, dim, g_params);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float beta1, float beta2,
                        float eps, float grad_scale, void* stream) {
    UPS_CHECK_ARG(p && g && m && v && count >= 0);
    if (count == 0) return UPS_OK;
    long long grid = (count + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)count, lr_t,
                       (const float*)nullptr, beta1, beta2, eps, grad_scale);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_adam_dev(float* p, const float* g, float* m, float* v, int64_t count, const float* lr_t_dev, float beta1,
                            float beta2, float eps, float grad_scale, void* stream) {
    UPS_CHECK_ARG(p && g && m && v && lr_t_dev && count >= 0);
    if (count == 0) return UPS_OK;
    long long grid = (count + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(adam_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)count, 0.f, lr_t_dev,
                       beta1, beta2, eps, grad_scale);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

// The step's state update (cub/code/SB_model48i/model.py:28-35, 829-834, 861-866, 890-909, 921-930) in ONE launch: seven EMAs and the
// two multipliers.  As ~30 scalar torch launches this was the last thing a step enqueued, with the GPU long drained behind it.  The
// products and sums are rounded one by one (no contraction), as the scalar torch expressions they replace were.
namespace {
__global__ void state_update_kernel(const float* __restrict__ stats, const float* __restrict__ old, float* __restrict__ out,
                                    float decay, float gain, int up_loa, float loa_lr, float loa_target, int up_lor, float lor_lr,
                                    float lor_target, float lor_min, float lor_max) {
    // every product and sum below is rounded on its own, as the torch expressions it replaces round them (the test holds it to them bit for
    // bit).  __fmul_rn / __fadd_rn do NOT stop hipcc's contraction (their bodies are compiled under the header's fp-contract=fast: the build
    // without packed fp32 forms fused decay * o + gain * val into v_fmac_f32, round 6); plain operators under this pragma are not fused.
#pragma clang fp contract(off)
    const int i = threadIdx.x;
    if (i >= 9) return;
    const float mim = stats[0], ind = stats[1], acc0 = stats[2], acc1 = stats[3], l0 = stats[4], l1 = stats[5];
    const float o = old[i];
    float v;
    if (i < 7) {
        const float val = i == 0 ? acc0 : i == 1 ? acc1 : i == 2 ? acc1 - acc0 : i == 3 ? l0 : i == 4 ? l1 : i == 5 ? mim : ind;
        const float a = decay * o, b = gain * val;
        v = a + b;
    } else if (i == 7) {
        const float d = loa_lr * (mim - loa_target);
        v = up_loa ? fmaxf(o + d, 0.f) : o;
    } else {
        const float d = lor_lr * (ind - lor_target);
        v = up_lor ? fminf(fmaxf(o + d, lor_min), lor_max) : o;
    }
    out[i] = v;
}
}  // namespace

extern "C" int ups_state_update(const float* stats, const float* old_state, float* new_state, float decay, float gain,
                                int32_t update_loa, float loa_lr, float loa_target, int32_t update_lor, float lor_lr,
                                float lor_target, float lor_min, float lor_max, void* stream) {
    UPS_CHECK_ARG(stats && old_state && new_state && lor_min <= lor_max);
    hipLaunchKernelGGL(state_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, old_state, new_state, decay, gain,
                       update_loa, loa_lr, loa_target, update_lor, lor_lr, lor_target, lor_min, lor_max);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_gauss_hm(const float* pts, const float* stddev, float* out, int32_t B, int32_t h, int32_t w, int32_t K,
                            void* stream) {
    UPS_CHECK_ARG(pts && stddev && out);
    hipLaunchKernelGGL(gauss_hm_kernel, dim3(ups_cdiv((long long)B * h * w * K, 256)), dim3(256), 0, (hipStream_t)stream, pts,
                       stddev, out, B, h, w, K);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
extern "C" int ups_gauss_hm3(const float* mu, const float* L, float* out, int32_t B, int32_t h, int32_t w, int32_t K,
                             void* stream) {
    UPS_CHECK_ARG(mu && L && out);
    hipLaunchKernelGGL(gauss_hm3_kernel, dim3(ups_cdiv((long long)B * h * w * K, 256)), dim3(256), 0, (hipStream_t)stream, mu, L,
                       out, B, h, w, K);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
