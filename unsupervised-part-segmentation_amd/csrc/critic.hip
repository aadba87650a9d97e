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
// (round 6, late: sixteen waves and 16-byte loads, four rows of a wave in flight at once.  Four waves walking 32 rows each with 2-byte
// loads and a reduction per row were 2B dependent round trips in a row: 80 us for 128 rows, three times per step)
template <typename T>
__global__ __launch_bounds__(1024) void critic_head_fwd_kernel(const T* __restrict__ hp, const T* __restrict__ ha, int B, int K, int ld,
                                                               float* __restrict__ logits, float* __restrict__ out) {
    extern __shared__ float lg[];                         // [2B]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool vec = sizeof(T) == 2 && (K & 7) == 0 && (ld & 7) == 0 && ((((unsigned long long)hp) | ((unsigned long long)ha)) & 15ull) == 0;
    bool done = false;
    if constexpr (sizeof(T) == 2) {
    if (vec) {
        done = true;
        const int nv = K >> 3;                            // 16-byte pieces per row
        for (int r0 = wv; r0 < 2 * B; r0 += 4 * nw) {     // rows r0, r0 + nw, r0 + 2 nw, r0 + 3 nw of this wave: their pieces requested together
            float sum[4] = {0.f, 0.f, 0.f, 0.f};
            for (int v = lane; v < nv; v += 64) {
                uint4 a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = min(r0 + q * nw, 2 * B - 1);          // (clamped: a row past the end is read and dropped)
                    a[q] = *(const uint4*)(hp + (long long)r * ld + 8 * v);
                    b[q] = *(const uint4*)(ha + (long long)r * ld + 8 * v);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned aw[4] = {a[q].x, a[q].y, a[q].z, a[q].w}, bw[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float a0, a1, b0, b1;
                        ups_unpack2<T>(aw[k], a0, a1); ups_unpack2<T>(bw[k], b0, b1);
                        sum[q] += a0 * b0; sum[q] += a1 * b1;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = wave_sum(sum[q]);
                const int r = r0 + q * nw;
                if (lane == 0 && r < 2 * B) { lg[r] = d; logits[r] = d; }
            }
        }
    }
    }
    if (!done) {
        for (int r = wv; r < 2 * B; r += nw) {
            const float d = row_dot(hp + (long long)r * ld, ha + (long long)r * ld, K, lane);
            if (lane == 0) { lg[r] = d; logits[r] = d; }
        }
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


// ---------------------------------------------------------------------------------------------------------------------------------
// The critics' towers as grouped launches (round 5).  discriminator_model (cub/code/SB_model48i/model.py:159-173) is two towers of
// nin -> 4 x residual_block(k = 1) -> nin on [2B, 512] rows; the three critics make six towers of six 1x1 layers that the generic
// convolution path ran as 72 GEMMs of 32 blocks + 65 split-K epilogues + 36 weight gradients + 36 bias sums per step, every one of
// them launch latency.  Here ONE launch runs the same layer of every tower (blockIdx.z = tower), and ONE launch takes every weight
// and bias gradient of every tower and layer.  Storage is the generic path's post-activation form: a layer's output is stored as
// lrelu(x) when an activated layer consumes it, the residual x is recovered from it (x = h > 0 ? h : h / slope), act' is read off
// the sign of the stored tensor.
//
// GEMM: out[m][n] = sum_k A[m][k] * W[k / 32][n][k % 32] (ups_weight_prep's blocked-K layout), fp32 accumulate.  No LDS: the
// operands are a few hundred KB that live in L2 and every lane's 16-byte pieces are the MFMA fragments as they lie in memory
// (weights = first operand: a lane's four accumulators are four consecutive channels of one row).  One wave = 16 rows x 32 channels.
struct TowerGemmArgs {
    const bf16* A[8]; const bf16* W[8]; const float* bias[8]; const bf16* sgn[8]; const bf16* res[8]; bf16* out[8];
    int lda[8], lds[8], ldr[8], ldo[8], K[8];
    int M, N;
    int res_self, out_act;          // forward: + the residual recovered from A (K == N); store lrelu(v)
    float slope;
};

__global__ __launch_bounds__(256) void tower_gemm_kernel(const TowerGemmArgs a) {
    const int tw = blockIdx.z;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p16 = lane & 15, q16 = lane >> 4;
    const int m = blockIdx.y * 16 + p16;
    const int n0 = blockIdx.x * 128 + wid * 32;
    const int mc = min(m, a.M - 1);
    const int N = a.N, nkc = a.K[tw] >> 5;
    const bf16* arow = a.A[tw] + (long long)mc * a.lda[tw] + q16 * 8;
    const bf16* wrow = a.W[tw] + (long long)(n0 + p16) * 32 + q16 * 8;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int kc = 0;
    for (; kc + 8 <= nkc; kc += 8) {            // 24 independent 16-byte loads in flight, then their 16 MFMAs
        bf16x8 af[8], w0[8], w1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            af[u] = *(const bf16x8*)(arow + (kc + u) * 32);
            w0[u] = *(const bf16x8*)(wrow + (long long)(kc + u) * N * 32);
            w1[u] = *(const bf16x8*)(wrow + (long long)(kc + u) * N * 32 + 16 * 32);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], af[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], af[u], acc[1], 0, 0, 0);
        }
    }
    for (; kc + 2 <= nkc; kc += 2) {
        bf16x8 af[2], w0[2], w1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            af[u] = *(const bf16x8*)(arow + (kc + u) * 32);
            w0[u] = *(const bf16x8*)(wrow + (long long)(kc + u) * N * 32);
            w1[u] = *(const bf16x8*)(wrow + (long long)(kc + u) * N * 32 + 16 * 32);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], af[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], af[u], acc[1], 0, 0, 0);
        }
    }
    for (; kc < nkc; ++kc) {
        const bf16x8 af = *(const bf16x8*)(arow + kc * 32);
        const bf16x8 w0 = *(const bf16x8*)(wrow + (long long)kc * N * 32);
        const bf16x8 w1 = *(const bf16x8*)(wrow + (long long)kc * N * 32 + 16 * 32);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af, acc[1], 0, 0, 0);
    }
    if (m >= a.M) return;
    const float inv = 1.f / a.slope;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 16 * j + 4 * q16;
        float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        if (a.bias[tw]) {               // (scalar loads: a bias is a 4-byte aligned slice of its key's flat parameter buffer)
            const float* bp = a.bias[tw] + n;
            v[0] += bp[0]; v[1] += bp[1]; v[2] += bp[2]; v[3] += bp[3];
        }
        if (a.sgn[tw]) {                // input gradient: act' off the sign of the stored forward input
            const uint2 sw = *(const uint2*)(a.sgn[tw] + (long long)m * a.lds[tw] + n);
            float s0, s1, s2, s3;
            ups_unpack2<bf16>(sw.x, s0, s1); ups_unpack2<bf16>(sw.y, s2, s3);
            v[0] *= s0 > 0.f ? 1.f : a.slope; v[1] *= s1 > 0.f ? 1.f : a.slope;
            v[2] *= s2 > 0.f ? 1.f : a.slope; v[3] *= s3 > 0.f ? 1.f : a.slope;
        }
        if (a.res[tw]) {                // input gradient of x + conv(act(x)): + g
            const uint2 rw = *(const uint2*)(a.res[tw] + (long long)m * a.ldr[tw] + n);
            float r0, r1, r2, r3;
            ups_unpack2<bf16>(rw.x, r0, r1); ups_unpack2<bf16>(rw.y, r2, r3);
            v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
        }
        if (a.res_self) {               // forward x + conv(act(x)) with x stored as act(x)
            const uint2 rw = *(const uint2*)(a.A[tw] + (long long)m * a.lda[tw] + n);
            float r0, r1, r2, r3;
            ups_unpack2<bf16>(rw.x, r0, r1); ups_unpack2<bf16>(rw.y, r2, r3);
            v[0] += r0 > 0.f ? r0 : r0 * inv; v[1] += r1 > 0.f ? r1 : r1 * inv;
            v[2] += r2 > 0.f ? r2 : r2 * inv; v[3] += r3 > 0.f ? r3 : r3 * inv;
        }
        if (a.out_act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ups_vmax(v[e], a.slope * v[e]);
        }
        *(uint2*)(a.out[tw] + (long long)m * a.ldo[tw] + n) = make_uint2(Chunk<bf16>::pk(v[0], v[1]), Chunk<bf16>::pk(v[2], v[3]));
    }
}

// Every weight and bias gradient of every tower and layer in one launch: item = (tower, layer), dW[k][n] = sum_m X[m][k] G[m][n]
// (HWIO of a 1x1 kernel), db[n] = sum_m G[m][n].  Block = 64 k x 64 n of one item, rows in chunks of 32 staged row-major in LDS;
// both MFMA operands are columns of those tiles: ds_read_b64_tr_b16 (guide T10; the same row assignment on both sides, so the
// permuted order of the reduction index does not matter).  No atomics: fixed summation order.
struct TowerWgItem { const bf16* X; const bf16* G; float* dW; float* db; int ldx, ldg, K, N; };
struct TowerWgArgs { TowerWgItem it[48]; int M; };
constexpr int TW_PITCH = 64 * 2 + 16;        // bytes per staged row (64 bf16 + 16: the four rows of a transposing read hit distinct banks)

__global__ __launch_bounds__(256) void tower_wgrad_kernel(const TowerWgArgs a) {
    typedef __attribute__((ext_vector_type(4))) short v4s;
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    __shared__ __attribute__((aligned(16))) unsigned char XS[32 * TW_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char GS[32 * TW_PITCH];
    const TowerWgItem& it = a.it[blockIdx.z];
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    if (k0 >= it.K || n0 >= it.N) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lg = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int srow = tid >> 3, spc = tid & 7;          // staging: thread -> (row of the chunk, 16-byte piece of the 64 columns)
    const bool kfull = k0 + 64 <= it.K;                 // (K = 64 * i always here; the check keeps a ragged K out of bounds)
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float bsum = 0.f;
    for (int mb = 0; mb < a.M; mb += 32) {
        const int m = mb + srow;
        uint4 xv = make_uint4(0u, 0u, 0u, 0u), gv = make_uint4(0u, 0u, 0u, 0u);
        if (m < a.M) {
            if (kfull || k0 + spc * 8 + 8 <= it.K) xv = *(const uint4*)(it.X + (long long)m * it.ldx + k0 + spc * 8);
            gv = *(const uint4*)(it.G + (long long)m * it.ldg + n0 + spc * 8);
        }
        __syncthreads();                                 // the previous chunk's reads are done
        *(uint4*)(XS + srow * TW_PITCH + spc * 16) = xv;
        *(uint4*)(GS + srow * TW_PITCH + spc * 16) = gv;
        __syncthreads();
        // rows (4 lg + q) and (16 + 4 lg + q) of the chunk: lane group lg holds reduction indices {4 lg .. 4 lg + 3, 16 + 4 lg .. }
        const v4s x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(XS + (4 * lg + q) * TW_PITCH + 2 * (16 * wid + 4 * pp)));
        const v4s x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(XS + (16 + 4 * lg + q) * TW_PITCH + 2 * (16 * wid + 4 * pp)));
        bf16x8 xf;
        __builtin_memcpy(&xf, &x0, 8);
        __builtin_memcpy((char*)&xf + 8, &x1, 8);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const v4s g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(GS + (4 * lg + q) * TW_PITCH + 2 * (16 * nb + 4 * pp)));
            const v4s g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(GS + (16 + 4 * lg + q) * TW_PITCH + 2 * (16 * nb + 4 * pp)));
            bf16x8 gf;
            __builtin_memcpy(&gf, &g0, 8);
            __builtin_memcpy((char*)&gf + 8, &g1, 8);
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, gf, acc[nb], 0, 0, 0);
        }
        if (blockIdx.y == 0 && tid < 64) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) bsum += (float)*(const bf16*)(GS + r * TW_PITCH + tid * 2);
        }
    }
    // lane (li, lg) holds dW rows k0 + 16 wid + 4 lg + e, column n0 + 16 nb + li
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + 16 * wid + 4 * lg + e;
            if (k < it.K) it.dW[(long long)k * it.N + n0 + 16 * nb + li] = acc[nb][e];
        }
    if (blockIdx.y == 0 && tid < 64 && it.db) it.db[n0 + tid] = bsum;
}

}  // namespace

extern "C" int ups_critic_head_fwd(const void* h_pi, const void* h_al, int32_t dtype, int32_t B, int32_t K, int32_t ld, float* logits,
                                   float* out4, void* stream) {
    UPS_CHECK_ARG(h_pi && h_al && logits && out4 && B > 0 && B <= 4096 && K > 0 && K <= ld);
    hipStream_t s = (hipStream_t)stream;
    const size_t shm = (size_t)2 * B * sizeof(float);
    if (dtype == UPS_F32) hipLaunchKernelGGL(critic_head_fwd_kernel<float>, dim3(1), dim3(1024), shm, s, (const float*)h_pi, (const float*)h_al, B, K, ld, logits, out4);
    else if (dtype == UPS_BF16) hipLaunchKernelGGL(critic_head_fwd_kernel<bf16>, dim3(1), dim3(1024), shm, s, (const bf16*)h_pi, (const bf16*)h_al, B, K, ld, logits, out4);
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

// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
bool tower_layers_ok(const ups_tower_layer* ly, int T, int L) {
    for (int i = 0; i < T * L; ++i) {
        const ups_tower_layer& l = ly[i];
        if (!l.w_fwd || !l.w_dgrad || l.k <= 0 || l.n <= 0 || l.k % 32 || l.n % 128) return false;
        if (((uintptr_t)l.w_fwd & 15) || ((uintptr_t)l.w_dgrad & 15) || ((uintptr_t)l.bias & 3) || ((uintptr_t)l.grad_w & 3) || ((uintptr_t)l.grad_b & 3)) return false;
        const int li = i % L;
        if (li > 0 && l.k != ly[i - 1].n) return false;             // the chain's widths
        if (li > 0 && li < L - 1 && l.k != l.n) return false;       // residual layers are square
    }
    return true;
}
}  // namespace

extern "C" int ups_towers_fwd(const ups_tower_layer* layers, int32_t T, int32_t L, const void* const* x0, const int32_t* ld0,
                              void* const* acts, int32_t M, float slope, void* stream) {
    UPS_CHECK_ARG(layers && x0 && ld0 && acts && T > 0 && T <= 8 && L >= 2 && L <= 6 && M > 0 && slope > 0.f && slope < 1.f);
    UPS_CHECK_ARG(tower_layers_ok(layers, T, L));
    hipStream_t s = (hipStream_t)stream;
    for (int l = 0; l < L; ++l) {
        TowerGemmArgs a = {};
        a.M = M; a.N = layers[l].n; a.slope = slope;
        a.res_self = (l > 0 && l < L - 1) ? 1 : 0;
        a.out_act = l < L - 1 ? 1 : 0;
        for (int t = 0; t < T; ++t) {
            const ups_tower_layer& ly = layers[t * L + l];
            UPS_CHECK_ARG(ly.n == a.N && acts[t * L + l] && (l > 0 || (x0[t] && ld0[t] >= ly.k && ld0[t] % 8 == 0)));
            UPS_CHECK_ARG(((uintptr_t)acts[t * L + l] & 15) == 0 && (l > 0 || ((uintptr_t)x0[t] & 15) == 0));      // 16-byte fragments
            a.A[t] = (const bf16*)(l == 0 ? x0[t] : acts[t * L + l - 1]);
            a.lda[t] = l == 0 ? ld0[t] : ly.k;
            a.W[t] = (const bf16*)ly.w_fwd; a.bias[t] = ly.bias; a.K[t] = ly.k;
            a.out[t] = (bf16*)acts[t * L + l]; a.ldo[t] = ly.n;
        }
        hipLaunchKernelGGL(tower_gemm_kernel, dim3(a.N / 128, (M + 15) / 16, T), dim3(256), 0, s, a);
    }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_towers_bwd(const ups_tower_layer* layers, int32_t T, int32_t L, const void* const* x0, const int32_t* ld0,
                              const void* const* acts, const void* const* g_out, void* const* g_ws, void* const* g_x0,
                              const int32_t* ldg0, int32_t want_wgrad, int32_t M, float slope, void* stream) {
    UPS_CHECK_ARG(layers && x0 && ld0 && acts && g_out && g_ws && T > 0 && T <= 8 && L >= 2 && L <= 6 && M > 0 && slope > 0.f && slope < 1.f);
    UPS_CHECK_ARG(tower_layers_ok(layers, T, L));
    hipStream_t s = (hipStream_t)stream;
    int act_t[8], na = 0;
    for (int t = 0; t < T; ++t) if (g_out[t]) act_t[na++] = t;
    if (!na) return UPS_OK;
    // gin[l] = the gradient w.r.t. layer l's (pre-storage-activation) output: g_out for l = L - 1, else g_ws[t * L + l]
    for (int l = L - 1; l >= 0; --l) {
        TowerGemmArgs a = {};
        a.M = M; a.slope = slope;
        int nt = 0;
        for (int i = 0; i < na; ++i) {
            const int t = act_t[i];
            const ups_tower_layer& ly = layers[t * L + l];
            const bf16* gin = (const bf16*)(l == L - 1 ? g_out[t] : g_ws[t * L + l]);
            UPS_CHECK_ARG(gin && ((uintptr_t)gin & 15) == 0);
            if (l == 0) {
                if (!g_x0 || !g_x0[t]) continue;
                UPS_CHECK_ARG(ldg0 && ldg0[t] >= ly.k && ly.k % 128 == 0);
                a.out[nt] = (bf16*)g_x0[t]; a.ldo[nt] = ldg0[t];
            } else {
                UPS_CHECK_ARG(g_ws[t * L + l - 1]);
                a.out[nt] = (bf16*)g_ws[t * L + l - 1]; a.ldo[nt] = ly.k;
                a.sgn[nt] = (const bf16*)acts[t * L + l - 1]; a.lds[nt] = ly.k;        // act' of the layer's stored input
                if (l < L - 1) { a.res[nt] = gin; a.ldr[nt] = ly.n; }                   // the residual stream's gradient
            }
            if (nt && a.N != ly.k) { ups_set_error("ups_towers_bwd: towers of different widths at layer %d", l); return UPS_E_ARG; }
            a.N = ly.k;                                   // the input gradient's "output channels" are the layer's inputs
            a.A[nt] = gin; a.lda[nt] = ly.n; a.K[nt] = ly.n;
            a.W[nt] = (const bf16*)ly.w_dgrad;
            ++nt;
        }
        if (nt) hipLaunchKernelGGL(tower_gemm_kernel, dim3(a.N / 128, (M + 15) / 16, nt), dim3(256), 0, s, a);
    }
    if (want_wgrad) {
        TowerWgArgs w = {};
        w.M = M;
        int ni = 0, kmax = 0, nmax = 0;
        for (int i = 0; i < na; ++i)
            for (int l = 0; l < L; ++l) {
                const int t = act_t[i];
                const ups_tower_layer& ly = layers[t * L + l];
                if (!ly.grad_w) continue;
                TowerWgItem& it = w.it[ni++];
                it.X = (const bf16*)(l == 0 ? x0[t] : acts[t * L + l - 1]); it.ldx = l == 0 ? ld0[t] : ly.k;
                it.G = (const bf16*)(l == L - 1 ? g_out[t] : g_ws[t * L + l]); it.ldg = ly.n;
                it.dW = ly.grad_w; it.db = ly.grad_b; it.K = ly.k; it.N = ly.n;
                kmax = ly.k > kmax ? ly.k : kmax; nmax = ly.n > nmax ? ly.n : nmax;
            }
        if (ni) hipLaunchKernelGGL(tower_wgrad_kernel, dim3(nmax / 64, (kmax + 63) / 64, ni), dim3(256), 0, s, w);
    }
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
