// Head of the separable MI critics (cub/code/SB_model48i/model.py:159-173 last line, 524-536, 800-834): the dot product of the two
// 512-d embeddings, the logistic losses of the joint / marginal pairs, the accuracy and the mean joint logit -- one launch forward,
// one backward (the torch formulation was ~25 element-wise launches per critic and step).  Latency-bound by construction: 2B rows of
// K = 512 values; ONE block, so that every reduction has a fixed order (bit-reproducible across runs and under HIP-graph replay).
#include "common.h"

namespace {

template <typename T>
__device__ inline float row_dot(const T* __restrict__ a, const T* __restrict__ b, int K, int lane) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += ld_as_float<T>(a + k) * ld_as_float<T>(b + k);
    return wave_sum(s);
}

__device__ inline float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ inline float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// logits[i] = <h_pi[i], h_al[i]>, rows [0, B) joint pairs, [B, 2B) marginal pairs.
// out[0] = 0.5 * (mean softplus(-joint) + mean softplus(marg))     logit_loss(real=True) / (real=False), model.py:524-529
// out[1] = (#joint > 0 + #marg < 0) / 2B                            model.py:821-826
// out[2] = mean joint                                               logit_constraint(real=False), model.py:532-536, 855
template <typename T>
__global__ __launch_bounds__(256) void critic_head_fwd_kernel(const T* __restrict__ hp, const T* __restrict__ ha, int B, int K, int ld,
                                                              float* __restrict__ logits, float* __restrict__ out) {
    extern __shared__ float lg[];                         // [2B]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv; r < 2 * B; r += 4) {
        const float d = row_dot(hp + (long long)r * ld, ha + (long long)r * ld, K, lane);
        if (lane == 0) { lg[r] = d; logits[r] = d; }
    }
    __syncthreads();
    if (wv == 0) {                                        // one wave, rows in a fixed order per lane, then a fixed shuffle tree
        float sj = 0.f, sm = 0.f, mj = 0.f, ok = 0.f;
        for (int r = lane; r < B; r += 64) {
            const float xj = lg[r], xm = lg[B + r];
            sj += softplus_f(-xj); sm += softplus_f(xm); mj += xj;
            ok += (xj > 0.f ? 1.f : 0.f) + (xm < 0.f ? 1.f : 0.f);
        }
        sj = wave_sum(sj); sm = wave_sum(sm); mj = wave_sum(mj); ok = wave_sum(ok);
        if (lane == 0) {
            out[0] = 0.5f * (sj / (float)B + sm / (float)B);
            out[1] = ok / (float)(2 * B);
            out[2] = mj / (float)B;
            out[3] = 0.f;
        }
    }
}

// d out[0] / d logit: joint -0.5 sigmoid(-x) / B, marginal +0.5 sigmoid(x) / B;  d out[2] / d joint = 1 / B.
// g_h_pi[i] = dlogit[i] * h_al[i], g_h_al[i] = dlogit[i] * h_pi[i]   (either output may be NULL)
template <typename T>
__global__ __launch_bounds__(256) void critic_head_bwd_kernel(const T* __restrict__ hp, const T* __restrict__ ha,
                                                              const float* __restrict__ logits, const float* __restrict__ g_loss,
                                                              const float* __restrict__ g_mim, int B, int K, int ld,
                                                              T* __restrict__ ghp, T* __restrict__ gha) {
    const int r = blockIdx.x;
    const float x = logits[r];
    const float gl = g_loss ? *g_loss : 0.f, gm = g_mim ? *g_mim : 0.f;
    const float dl = r < B ? (-0.5f * sigmoid_f(-x) * gl + gm) / (float)B : (0.5f * sigmoid_f(x) * gl) / (float)B;
    for (int k = threadIdx.x; k < ld; k += 256) {
        const float a = k < K ? ld_as_float<T>(hp + (long long)r * ld + k) : 0.f;
        const float b = k < K ? ld_as_float<T>(ha + (long long)r * ld + k) : 0.f;
        if (ghp) st_from_float<T>(ghp + (long long)r * ld + k, dl * b);
        if (gha) st_from_float<T>(gha + (long long)r * ld + k, dl * a);
    }
}

}  // namespace

extern "C" int ups_critic_head_fwd(const void* h_pi, const void* h_al, int32_t dtype, int32_t B, int32_t K, int32_t ld, float* logits,
                                   float* out4, void* stream) {
    UPS_CHECK_ARG(h_pi && h_al && logits && out4 && B > 0 && B <= 4096 && K > 0 && K <= ld);
    hipStream_t s = (hipStream_t)stream;
    const size_t shm = (size_t)2 * B * sizeof(float);
    if (dtype == UPS_F32) hipLaunchKernelGGL(critic_head_fwd_kernel<float>, dim3(1), dim3(256), shm, s, (const float*)h_pi, (const float*)h_al, B, K, ld, logits, out4);
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(critic_head_fwd_kernel<bf16>, dim3(1), dim3(256), shm, s, (const bf16*)h_pi, (const bf16*)h_al, B, K, ld, logits, out4);
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_critic_head_bwd(const void* h_pi, const void* h_al, const float* logits, const float* g_loss, const float* g_mim,
                                   int32_t dtype, int32_t B, int32_t K, int32_t ld, void* g_h_pi, void* g_h_al, void* stream) {
    UPS_CHECK_ARG(h_pi && h_al && logits && (g_h_pi || g_h_al) && B > 0 && K > 0 && K <= ld);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32) hipLaunchKernelGGL(critic_head_bwd_kernel<float>, dim3(2 * B), dim3(256), 0, s, (const float*)h_pi, (const float*)h_al, logits, g_loss, g_mim, B, K, ld, (float*)g_h_pi, (float*)g_h_al);
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(critic_head_bwd_kernel<bf16>, dim3(2 * B), dim3(256), 0, s, (const bf16*)h_pi, (const bf16*)h_al, logits, g_loss, g_mim, B, K, ld, (bf16*)g_h_pi, (bf16*)g_h_al);
    else { ups_set_error("bad dtype %d", (int)dtype); return UPS_E_ARG; }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
