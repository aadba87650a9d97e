// Weight (and bias) gradient of the gather convolution:
//     dV[tap][ci][co] = sum_pix act(in)[src(pix,tap)][ci] * dout[pix][co],   db[co] = sum_pix dout[pix][co]
// (gradient of tf.nn.conv2d + bias w.r.t. V and b, cub/code/nn.py:661-663).
//
// GEMM view: rows = "virtual channels" v = tap*ci + c (taps are packed into the row tile, so a 128-row tile of a
// thin layer covers several taps and dout is re-read ceil(ntaps*ci/128) times instead of ntaps times),
// cols = output channels, K = lattice points (pixels) -> split-K over pixels into fp32 slabs + a deterministic
// slab reduction (no float atomics: bitwise reproducible).
// Both operands are pixel-major in HBM (NHWC), i.e. K is the slow dimension of both: the tiles are staged
// row-major [pixel][channel] in LDS and the MFMA fragments are fetched with the gfx950 transposing read
// ds_read_b64_tr_b16 (bf16) / plain ds_read_b32 (f32 32x32x2 needs one element per lane).
// LDS row stride == 64 (mod 128) bytes keeps the transposed reads of a 32-lane half on distinct banks.
// The blocks of row-tile 0 also reduce their dout tiles over pixels -> bias gradient (no extra pass over dout).
#include "common.h"

namespace {

struct WgK {
    int n, hi, wi, ci, ldi, ci_log, cin_v, ho, wo, co, ldo, in_sy, in_sx, ntaps, act_in, want_bias, in_f16;
    float act_slope;
    unsigned long long tap_off, tap_wi;
    const void* in; const void* dout; float* ws;
    float* bias_direct;      // one split: ws is the gradient itself (same offsets as a slab's weight part), the bias goes here
};

__device__ inline int wtap_dy(unsigned long long off, int t) { return (int)((off >> (4 * t + 2)) & 3) - 1; }
__device__ inline int wtap_dx(unsigned long long off, int t) { return (int)((off >> (4 * t)) & 3) - 1; }
__device__ inline int wtap_w(unsigned long long wi, int t) { return (int)((wi >> (4 * t)) & 15); }

constexpr int lds_stride(int row_bytes) {  // smallest stride >= row_bytes with stride % 128 == 64
    return ((row_bytes + 63) / 128) * 128 + 64;
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// fragment for v_mfma_f32_32x32x16_bf16 from a row-major LDS tile M[k][c]:
// element j of lane l = M[k0 + 8*(l>>5) + j][c0 + (l&31)]
__device__ inline bf16x8 tr_frag(const unsigned char* tile, int rs, int k0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int cb = c0 + 16 * (g & 1), q = i >> 2, pp = i & 3;
    const unsigned char* a0 = tile + (k0 + 8 * h + q) * rs + (cb + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * rs));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

// slab layout per split: [ntaps*cin_v*co weight floats][co bias floats]
template <typename T, int BMW, int BNW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK p, const int M, const int cit, const int cot,
                                                         const int chunks_total, const int chunks_per) {
    constexpr int EPC = Chunk<T>::N;
    constexpr int PK = 4 * EPC;               // pixels per K-chunk: 32 (bf16) / 16 (f32)
    constexpr int RSA = lds_stride(BMW * (int)sizeof(T));
    constexpr int RSB = lds_stride(BNW * (int)sizeof(T));
    constexpr int NBM = BMW / 32, NBN = BNW / 32;
    constexpr int WM = NBM >= 2 ? ((NBN == 1 && NBM == 4) ? 4 : 2) : 1;
    constexpr int WN = NBN >= 2 ? ((NBM == 1 && NBN == 4) ? 4 : 2) : 1;
    constexpr int TM = NBM / WM, TN = NBN / WN;
    constexpr int ACTIVE = WM * WN;            // waves that own MFMA blocks (<= 4)
    constexpr int CPA = BMW / EPC;             // 16-byte chunks per X row
    constexpr int CPB = BNW / EPC;
    constexpr int NA = (PK * CPA + 255) / 256; // X chunks per thread
    constexpr int NB = (PK * CPB + 255) / 256;
    constexpr int BPARTS = 256 / BNW;          // pixel groups for the bias column sums

    __shared__ __attribute__((aligned(16))) unsigned char smem[PK * (RSA + RSB)];
    unsigned char* Xs = smem;
    unsigned char* Ds = smem + PK * RSA;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float act_ns = ups_slope_eff(p.act_in, p.act_slope);   // branch-free activation-on-load
    const int cot_i = blockIdx.x % cot, cit_i = blockIdx.x / cot;
    const int split = blockIdx.y;
    const int wm = wid / WN, wn = wid % WN;

    const T* __restrict__ in = (const T*)p.in;
    const T* __restrict__ dout = (const T*)p.dout;
    const int hw_o = p.ho * p.wo;
    const int V = p.ntaps * p.ci;
    const int v0 = cit_i * BMW, co0 = cot_i * BNW;
    const bool do_bias = p.want_bias && cit_i == 0;

    const int c_begin = split * chunks_per;
    const int c_end = min(chunks_total, c_begin + chunks_per);

    // loop-invariant decode of this thread's X items: virtual channel -> (tap offset, channel)
    int xrow0 = 0, xrow1 = 0, xc0 = -1, xc1 = -1, xdy0 = 0, xdx0 = 0, xdy1 = 0, xdx1 = 0;
    {
        auto dec = [&](int idx, int& row, int& c, int& dy, int& dx) {
            c = -1;
            if (idx < PK * CPA) {
                row = idx / CPA;
                const int v = v0 + (idx - row * CPA) * EPC;
                if (v < V) {
                    const int tap = v / p.ci;
                    c = v - tap * p.ci;
                    dy = wtap_dy(p.tap_off, tap); dx = wtap_dx(p.tap_off, tap);
                }
            }
        };
        dec(tid, xrow0, xc0, xdy0, xdx0);
        if (NA > 1) dec(tid + 256, xrow1, xc1, xdy1, xdx1);
    }

    uint4 xa0, xa1, db0, db1;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto load_x = [&](int chunk_idx, int row, int c, int dy, int dx) -> uint4 {
        uint4 v = zero4;
        const int m = chunk_idx * PK + row;
        if (c >= 0 && m < M) {
            const int img = m / hw_o, rem = m - img * hw_o;
            const int i = rem / p.wo, j = rem - i * p.wo;
            const int y = i * p.in_sy + dy, x = j * p.in_sx + dx;
            if ((unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi)
                v = *(const uint4*)(in + (((long long)img * p.hi + y) * p.wi + x) * p.ldi + c);
        }
        return v;
    };
    auto load_d = [&](int chunk_idx, int idx) -> uint4 {
        uint4 v = zero4;
        if (idx < PK * CPB) {
            const int row = idx / CPB, cc = idx - row * CPB;
            const int m = chunk_idx * PK + row;
            const int ch = co0 + cc * EPC;
            if (m < M && ch < p.ldo) v = *(const uint4*)(dout + (long long)m * p.ldo + ch);
        }
        return v;
    };
    auto act_u4 = [&](uint4 u) -> uint4 {
        if constexpr (__is_same(T, bf16)) {
            if (p.in_f16) return ups_act_chunk_f16_to_bf16(u, act_ns, p.act_in != UPS_ACT_NONE);
        }
        if (p.act_in != UPS_ACT_NONE) {
            u = ups_act_chunk(u, act_ns, (T*)nullptr);
        }
        return u;
    };
    auto load_chunk = [&](int chunk_idx) {
        xa0 = load_x(chunk_idx, xrow0, xc0, xdy0, xdx0);
        if (NA > 1) xa1 = load_x(chunk_idx, xrow1, xc1, xdy1, xdx1);
        db0 = load_d(chunk_idx, tid);
        if (NB > 1) db1 = load_d(chunk_idx, tid + 256);
    };
    auto stage_chunk = [&]() {
        if (tid < PK * CPA) { const int row = tid / CPA, cc = tid - row * CPA; *(uint4*)(Xs + row * RSA + cc * 16) = act_u4(xa0); }
        if (NA > 1) { const int idx = tid + 256, row = idx / CPA, cc = idx - row * CPA; *(uint4*)(Xs + row * RSA + cc * 16) = act_u4(xa1); }
        if (tid < PK * CPB) { const int row = tid / CPB, cc = tid - row * CPB; *(uint4*)(Ds + row * RSB + cc * 16) = db0; }
        if (NB > 1) { const int idx = tid + 256, row = idx / CPB, cc = idx - row * CPB; *(uint4*)(Ds + row * RSB + cc * 16) = db1; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;
    const int bcol = tid % BNW, bpart = tid / BNW;

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        stage_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < PK / BPARTS; ++k)
                bsum += ld_as_float<T>((const T*)(Ds + (bpart * (PK / BPARTS) + k) * RSB) + bcol);
        }
        if (wid < ACTIVE) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 a[TM], bb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = tr_frag(Xs, RSA, ks * 16, (wm * TM + i) * 32, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bb[j] = tr_frag(Ds, RSB, ks * 16, (wn * TN + j) * 32, lane);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
                }
            } else {
                const int r = lane & 31, h = lane >> 5;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    float a[TM], bb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *(const float*)(Xs + (8 * h + kk) * RSA + ((wm * TM + i) * 32 + r) * 4);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bb[j] = *(const float*)(Ds + (8 * h + kk) * RSB + ((wn * TN + j) * 32 + r) * 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    const long long slab_sz = (long long)p.ntaps * p.cin_v * p.co + p.co;
    float* slab = p.ws + (long long)split * slab_sz;
    if (wid < ACTIVE) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = co0 + (wn * TN + tn) * 32 + (lane & 31);
            if (col >= p.co) continue;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int v = v0 + (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (v < V) {
                        const int tap = v / p.ci, cch = v - tap * p.ci;
                        if (cch < p.ci_log)
                            slab[((long long)wtap_w(p.tap_wi, tap) * p.cin_v + cch) * p.co + col] = acc[tm][tn][e];
                    }
                }
        }
    }
    if (do_bias) {
        float* red = (float*)smem;            // safe: every wave passed the loop's final barrier
        red[tid] = bsum;
        __syncthreads();
        if (tid < BNW && co0 + tid < p.co) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < BPARTS; ++k) s += red[k * BNW + tid];
            if (p.bias_direct) p.bias_direct[co0 + tid] = s;
            else slab[(long long)p.ntaps * p.cin_v * p.co + co0 + tid] = s;
        }
    }
}

// grad[tap][row < ci_log][co] = sum_s slab[s][...] ; grad_bias[co] = sum_s slab[s][bias part]
// 256 threads = 32 consecutive outputs x 8 slab groups: a thread sums every 8th slab (4 loads in flight), the groups are
// combined through LDS in a fixed order.  (One thread per output walking all the slabs -- up to 256 dependent round trips
// on grids of a few dozen blocks for the thin layers -- cost 3.6 ms of kernel time per step.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ grad,
                                                           float* __restrict__ gbias, int splitk, int ntaps_w, int cin_v,
                                                           int ci_log, int co, long long slab) {
    __shared__ float red[8][32];
    const long long nw = (long long)ntaps_w * ci_log * co;
    const long long total = nw + (gbias ? co : 0);
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (long long base = (long long)blockIdx.x * 32; base < total; base += (long long)gridDim.x * 32) {
        const long long idx = base + o;
        const bool ok = idx < total;
        long long off = 0;
        float* dst = nullptr;
        if (ok) {
            if (idx < nw) {
                const int c = (int)(idx % co);
                const long long tr = idx / co;
                const int row = (int)(tr % ci_log), tap = (int)(tr / ci_log);
                off = ((long long)tap * cin_v + row) * co + c;
                dst = grad + off;
            } else {
                off = (long long)ntaps_w * cin_v * co + (idx - nw);
                dst = gbias + (idx - nw);
            }
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (ok) {
            int k = g;
            for (; k + 24 < splitk; k += 32) {
                s0 += ws[(long long)k * slab + off]; s1 += ws[(long long)(k + 8) * slab + off];
                s2 += ws[(long long)(k + 16) * slab + off]; s3 += ws[(long long)(k + 24) * slab + off];
            }
            for (; k < splitk; k += 8) s0 += ws[(long long)k * slab + off];
        }
        red[g][o] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && ok) {
            float t = red[0][o];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += red[q][o];
            *dst = t;
        }
        __syncthreads();
    }
}

template <typename T, int BMW, int BNW>
void launch_tile(const WgK& k, int M, int splitk, hipStream_t s) {
    constexpr int PK = 4 * Chunk<T>::N;
    const int cit = ups_cdiv(k.ntaps * k.ci, BMW), cot = ups_cdiv(k.co, BNW);
    const int chunks_total = ups_cdiv(M, PK);
    const int chunks_per = ups_cdiv(chunks_total, splitk);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, BMW, BNW>), dim3(cit * cot, splitk), dim3(256), 0, s, k, M, cit, cot,
                       chunks_total, chunks_per);
}

template <typename T, int BMW>
void launch_n(const WgK& k, int M, int splitk, hipStream_t s) {
    if (k.co > 64) launch_tile<T, BMW, 128>(k, M, splitk, s);
    else if (k.co > 32) launch_tile<T, BMW, 64>(k, M, splitk, s);
    else launch_tile<T, BMW, 32>(k, M, splitk, s);
}

int tile_of(int c) { return c > 64 ? 128 : (c > 32 ? 64 : 32); }

template <typename T>
void launch_m(const WgK& k, int M, int splitk, hipStream_t s) {
    const int bm = tile_of(k.ntaps * k.ci);
    if (bm == 128) launch_n<T, 128>(k, M, splitk, s);
    else if (bm == 64) launch_n<T, 64>(k, M, splitk, s);
    else launch_n<T, 32>(k, M, splitk, s);
}

int plan_splitk(const ups_wgrad_desc* d) {
    const long long M = (long long)d->n * d->ho * d->wo;
    const int pk = d->dtype == UPS_BF16 ? 32 : 16;
    const int v = d->ntaps * d->ci;
    const int tiles = ups_cdiv(v, tile_of(v)) * ups_cdiv(d->co, tile_of(d->co));
    const int chunks = ups_cdiv(M, pk);
    int sk = ups_cdiv(2048, tiles);            // ~8 blocks per CU
    const int max_by_work = chunks / 8 > 0 ? chunks / 8 : 1;  // at least 8 chunks per block
    if (sk > max_by_work) sk = max_by_work;
    if (sk > 512) sk = 512;
    if (sk < 1) sk = 1;
    return sk;
}

}  // namespace

int ups_wgrad3x3_plan(const ups_wgrad_desc* d, int* splitk, int* slabs);   // conv_wgrad3x3.hip
int ups_wgrad3x3_run(const ups_wgrad_desc* d, hipStream_t s);
int ups_wgrad3x3_f8_plan(const ups_wgrad_desc* d, int* splitk, int* slabs);   // conv_wgrad3x3_f8.hip
int ups_wgrad3x3_f8_run(const ups_wgrad_desc* d, hipStream_t s);

extern "C" int ups_conv_wgrad_plan(const ups_wgrad_desc* d, int32_t* splitk, size_t* workspace_bytes) {
    UPS_CHECK_ARG(d && splitk && workspace_bytes);
    UPS_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= 9 && d->ci > 0 && d->co > 0);
    int sk3 = 0, slabs3 = 0;
    if (ups_wgrad3x3_f8_plan(d, &sk3, &slabs3) == 0 ||    // e5m2 copy of dout given, wide 3x3 / stride-1: the fp8 kernel
        ups_wgrad3x3_plan(d, &sk3, &slabs3) == 0) {      // bf16 3x3/stride-1: patch-tiled kernel
        *splitk = sk3;
        *workspace_bytes = (size_t)slabs3 * ((size_t)d->ntaps * d->cin_v * d->co + d->co) * sizeof(float);
        return UPS_OK;
    }
    const int sk = plan_splitk(d);
    *splitk = sk;
    *workspace_bytes = (size_t)sk * ((size_t)d->ntaps * d->cin_v * d->co + d->co) * sizeof(float);
    return UPS_OK;
}

extern "C" int ups_conv_wgrad(const ups_wgrad_desc* d, void* stream) {
    UPS_CHECK_ARG(d != nullptr);
    UPS_CHECK_ARG(d->dtype == UPS_F32 || d->dtype == UPS_BF16);
    UPS_CHECK_ARG(d->in && d->dout && d->grad && d->workspace);
    UPS_CHECK_ARG(d->ci > 0 && d->ci % 8 == 0 && d->ldi % 8 == 0 && d->ci <= d->ldi && d->ldo % 8 == 0);
    UPS_CHECK_ARG(d->ci_log >= 1 && d->ci_log <= d->ci && d->cin_v >= d->ci_log);
    UPS_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= 9 && d->splitk >= 1);
    UPS_CHECK_ARG(d->act_slope >= 0.f && d->act_slope <= 1.f);    // activation-on-load is max(x, slope * x)
    UPS_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->dout & 15) == 0);
    const long long M = (long long)d->n * d->ho * d->wo;
    UPS_CHECK_ARG(M > 0 && M <= 0x7fffffffLL);
    WgK k;
    k.n = d->n; k.hi = d->hi; k.wi = d->wi; k.ci = d->ci; k.ldi = d->ldi; k.ci_log = d->ci_log; k.cin_v = d->cin_v;
    k.ho = d->ho; k.wo = d->wo; k.co = d->co; k.ldo = d->ldo; k.in_sy = d->in_sy; k.in_sx = d->in_sx;
    k.ntaps = d->ntaps; k.act_in = d->act_in; k.act_slope = d->act_slope; k.want_bias = d->grad_bias != nullptr;
    k.in_f16 = d->in_f16;
    UPS_CHECK_ARG(!d->in_f16 || (d->dtype == UPS_BF16 && !d->mask_bits));
    k.in = d->in; k.dout = d->dout; k.ws = d->workspace; k.bias_direct = nullptr;
    k.tap_off = 0; k.tap_wi = 0;
    int max_tw = 0;
    for (int t = 0; t < d->ntaps; ++t) {
        UPS_CHECK_ARG(d->tap_dy[t] >= -1 && d->tap_dy[t] <= 2 && d->tap_dx[t] >= -1 && d->tap_dx[t] <= 2);
        UPS_CHECK_ARG(d->tap_w[t] >= 0 && d->tap_w[t] <= 15);
        k.tap_off |= (unsigned long long)(((d->tap_dy[t] + 1) << 2) | (d->tap_dx[t] + 1)) << (4 * t);
        k.tap_wi |= (unsigned long long)d->tap_w[t] << (4 * t);
        if (d->tap_w[t] > max_tw) max_tw = d->tap_w[t];
    }
    UPS_CHECK_ARG(max_tw < d->ntaps);  // slab holds ntaps slices
    hipStream_t s = (hipStream_t)stream;
    int nslabs = d->splitk;
    int sk3 = 0, slabs3 = 0;
    if (d->mask_bits) {
        UPS_CHECK_ARG(d->mask_batch > 0 && d->n % d->mask_batch == 0 && d->n / d->mask_batch <= 32);
        if (ups_wgrad3x3_plan(d, &sk3, &slabs3) != 0) {
            ups_set_error("ups_conv_wgrad: the part-masked form needs the bf16 3x3 / stride-1 patch kernel (16-aligned images)");
            return UPS_E_UNSUPPORTED;
        }
    }
    if (ups_wgrad3x3_f8_plan(d, &sk3, &slabs3) == 0) {
        UPS_CHECK_ARG(d->splitk == sk3 && d->dout_f8_scale && d->in_f8_scale && ((uintptr_t)d->dout_f8 & 15) == 0);
        if (ups_wgrad3x3_f8_run(d, s) != UPS_OK) { ups_set_error("ups_conv_wgrad: fp8 kernel launch setup failed"); return UPS_E_LAUNCH; }
        nslabs = slabs3;
    } else if (ups_wgrad3x3_plan(d, &sk3, &slabs3) == 0) {
        UPS_CHECK_ARG(d->splitk == sk3);
        if (ups_wgrad3x3_run(d, s) != UPS_OK) { ups_set_error("ups_conv_wgrad: patch kernel launch setup failed"); return UPS_E_LAUNCH; }
        nslabs = slabs3;
    } else {
        // a single split writes exactly the elements the reduction would (rows < ci_log of every tap, the bias): straight into the
        // gradient, no slab and no second launch (the critics' 30 dense layers; UPS_WGRAD_DIRECT=0: off)
        static int direct = -1;
        if (direct < 0) { const char* e = getenv("UPS_WGRAD_DIRECT"); direct = (e && e[0] == '0') ? 0 : 1; }
        const bool one = direct && d->splitk == 1;
        if (one) { k.ws = d->grad; k.bias_direct = d->grad_bias; }
        if (d->dtype == UPS_F32) launch_m<float>(k, (int)M, d->splitk, s);
        else launch_m<bf16>(k, (int)M, d->splitk, s);
        UPS_LAUNCH_CHECK();
        if (one) return UPS_OK;
    }
    UPS_LAUNCH_CHECK();
    const long long slab = (long long)d->ntaps * d->cin_v * d->co + d->co;
    const long long total = (long long)d->ntaps * d->ci_log * d->co + d->co;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ups_cdiv(total, 32) > 8192 ? 8192 : ups_cdiv(total, 32)), dim3(256), 0,
                       s, d->workspace, d->grad, d->grad_bias, nslabs, d->ntaps, d->cin_v, d->ci_log, d->co, slab);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
