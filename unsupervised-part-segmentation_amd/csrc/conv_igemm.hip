// Implicit-GEMM gather convolution on the CDNA4 matrix cores (forward + dgrad).
//
// Replaces tf.nn.conv2d / its input gradient as used by nn.conv2d (cub/code/nn.py:617-711),
// nin (nn.py:811-813), downsample (nn.py:816-817) and residual_block (nn.py:1042-1056).
//
// GEMM view: M = n*ho*wo lattice points (rows), N = output channels, K = taps x ci.
//   A[m][k] = act(in[src(m, tap)][k])   gathered on the fly (never materialised: no im2col buffer)
//   B[k][c] = w[tap][c][k]               (k contiguous per output channel)
// Block = 256 threads = 4 waves, tile 128 x BN, K-chunk = 64 bytes per row (32 bf16 / 16 f32).
// Staging is through registers (global_load_dwordx4 -> fused activation -> ds_write_b128) because the
// activation-on-load and the zero padding of out-of-image taps need the data in VGPRs; the next chunk's
// global loads are issued before the MFMAs of the current one (issue-early / write-late).
// LDS rows are padded 64 -> 80 bytes so the 16-lane groups of ds_read_b128 fall on distinct 16-B slots.
// MFMA: v_mfma_f32_32x32x16_bf16, or v_mfma_f32_32x32x2_f32 (exact fp32) for the parity mode.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int RS = 80;  // LDS row stride in bytes (64 data + 16 pad)

template <typename T> struct Mma;

template <> struct Mma<bf16> {
    // one 64-byte K-chunk = 32 bf16 = two 32x32x16 steps
    template <int TM, int TN>
    __device__ static inline void chunk(const unsigned char* a_base, const unsigned char* b_base, int lane,
                                        f32x16 (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(a_base + (i * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(b_base + (j * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <> struct Mma<f16> {
    // as bf16 on v_mfma_f32_32x32x16_f16 (same rate, 10 mantissa bits)
    template <int TM, int TN>
    __device__ static inline void chunk(const unsigned char* a_base, const unsigned char* b_base, int lane,
                                        f32x16 (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f16x8*)(a_base + (i * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f16x8*)(b_base + (j * 32 + r) * RS + ks * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <> struct Mma<float> {
    // one 64-byte K-chunk = 16 floats; lane half h owns k in [8h, 8h+8): 8 x (32x32x2) steps
    template <int TM, int TN>
    __device__ static inline void chunk(const unsigned char* a_base, const unsigned char* b_base, int lane,
                                        f32x16 (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
        f32x4 a[TM][2], b[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a[i][0] = *(const f32x4*)(a_base + (i * 32 + r) * RS + h * 32);
            a[i][1] = *(const f32x4*)(a_base + (i * 32 + r) * RS + h * 32 + 16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            b[j][0] = *(const f32x4*)(b_base + (j * 32 + r) * RS + h * 32);
            b[j][1] = *(const f32x4*)(b_base + (j * 32 + r) * RS + h * 32 + 16);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk >> 2][kk & 3], b[j][kk >> 2][kk & 3],
                                                                     acc[i][j], 0, 0, 0);
    }
};

// taps packed 4 bits each (no dynamically indexed kernarg arrays -> no scratch):
//   tap_off: (dy+1)<<2 | (dx+1), dy,dx in [-1,2];   tap_wi: weight slice index
__device__ inline int tap_dy_of(unsigned long long off, int t) { return (int)((off >> (4 * t + 2)) & 3) - 1; }
__device__ inline int tap_dx_of(unsigned long long off, int t) { return (int)((off >> (4 * t)) & 3) - 1; }
__device__ inline int tap_w_of(unsigned long long wi, int t) { return (int)((wi >> (4 * t)) & 15); }

// ups_conv_desc without the tap arrays
struct ConvK {
    int n, hi, wi, ci, ldi, ho, wo, co, co_fill, ldo, out_h, out_w, out_sy, out_sx, out_oy, out_ox, in_sy, in_sx;
    int ntaps, kh, kw, act_in, out_f32, dact_kind, ldr, ldd, out_act, res_act;
    float act_slope;
    unsigned long long tap_off, tap_wi;
    const void* in; const void* w; void* out;
    const float* bias; const float* coord_tab; const void* res; const void* dact;
    float* ws; int splits, stages_per_split, ldw;   // split-K: partial [split][M][ldw] fp32 (ldw = ntn * BN)
};

// CPS = K-chunks per pipeline stage: skinny problems (few blocks, long K loops) are bound by one exposed memory
// latency per barrier, so they stage 2-4 chunks per barrier.
template <typename T, int BN, int CPS>
__global__ __launch_bounds__(256, (CPS == 1 ? 3 : 2)) void conv_igemm_kernel(const ConvK p, const int M, const int ntn,
                                                         const int kchunks) {
    constexpr int EPC = Chunk<T>::N;
    constexpr int BK = 4 * EPC;
    constexpr int WN = (BN == 32) ? 1 : 2;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int BROWS = (BN + 63) / 64;  // B rows staged per thread (BN=32: only threads < 128)

    constexpr int SUB = (BM + BN) * RS;         // one chunk: A tile + B tile
    constexpr int STAGE = CPS * SUB;            // one LDS stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE + BM * 20
    long long* rowpix = (long long*)(smem + 2 * STAGE);
    int* rowaux = (int*)(rowpix + BM);  // [BM][3] = cls, j, i

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float act_ns = ups_slope_eff(p.act_in, p.act_slope);   // branch-free activation-on-load
    const float dact_ns = ups_slope_eff(p.dact_kind, p.act_slope); // act'(x) = x > 0 ? 1 : dact_ns (only used when dact != NULL)
    const int nt = blockIdx.x % ntn, mt = blockIdx.x / ntn;
    const int wm = wid / WN, wn = wid % WN;

    const T* __restrict__ in = (const T*)p.in;
    const T* __restrict__ w = (const T*)p.w;
    const int hw_o = p.ho * p.wo;

    // ---- per-thread staging coordinates: rows r0 and r0+64, 16-byte chunk `chunk` of each row
    const int chunk = tid & 3, r0 = tid >> 2;
    int iy0[2], ix0[2];
    long long ibase[2];
    bool rv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int m = mt * BM + r0 + 64 * q;
        rv[q] = m < M;
        const int mm = rv[q] ? m : 0;
        const int img = mm / hw_o, rem = mm - img * hw_o;
        const int i = rem / p.wo, j = rem - i * p.wo;
        iy0[q] = i * p.in_sy;
        ix0[q] = j * p.in_sx;
        ibase[q] = (long long)img * p.hi * p.wi;
    }
    const bool b_thread = (BN >= 64) || (tid < 128);

    // two stages in flight (prefetch distance 2); every lambda is force-inlined and every index static after
    // unrolling, so the register arrays stay in VGPRs
    struct RSet { uint4 a0[CPS], a1[CPS], b0[CPS], b1[CPS]; } s0, s1;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const int iy00 = iy0[0], iy01 = iy0[1], ix00 = ix0[0], ix01 = ix0[1];
    const long long ib0 = ibase[0], ib1 = ibase[1];
    const bool rv0 = rv[0], rv1 = rv[1];
    const int total = p.ntaps * kchunks;

    auto load_a = [&](bool rvq, int iy, int ix, long long ib, int dy, int dx, int koff, bool kok) __attribute__((always_inline)) -> uint4 {
        const int y = iy + dy, x = ix + dx;
        const bool ok = rvq && kok && (unsigned)y < (unsigned)p.hi && (unsigned)x < (unsigned)p.wi;
        uint4 v = zero4;
        if (ok) v = *(const uint4*)(in + ((ib + (long long)y * p.wi + x) * p.ldi + koff));
        return v;
    };
    // weights are stored blocked-K: [tap][k-chunk][row][BK] (conv_aux.hip) -> a tile is one contiguous range
    auto load_b = [&](const T* wt, int c, bool live) __attribute__((always_inline)) -> uint4 {
        uint4 v = zero4;
        if (live && b_thread && c < p.co) v = *(const uint4*)(wt + (long long)c * BK + chunk * EPC);
        return v;
    };
    auto load_stage = [&](RSet& q, int sidx) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPS; ++u) {
            const int c = sidx * CPS + u;
            const bool live = c < total;
            const int cs = live ? c : 0;
            const int t = cs / kchunks, kc = cs - t * kchunks;
            const int koff = kc * BK + chunk * EPC;
            const bool kok = live && koff < p.ci;
            const int dy = tap_dy_of(p.tap_off, t), dx = tap_dx_of(p.tap_off, t);
            q.a0[u] = load_a(rv0, iy00, ix00, ib0, dy, dx, koff, kok);
            q.a1[u] = load_a(rv1, iy01, ix01, ib1, dy, dx, koff, kok);
            const T* wt = w + ((long long)tap_w_of(p.tap_wi, t) * kchunks + kc) * p.co * BK;
            q.b0[u] = load_b(wt, nt * BN + r0, live);
            if (BROWS > 1) q.b1[u] = load_b(wt, nt * BN + r0 + 64, live);
        }
    };
    auto act_u4 = [&](uint4 u) __attribute__((always_inline)) -> uint4 {
        if (p.act_in != UPS_ACT_NONE) {
            u = ups_act_chunk(u, act_ns, (T*)nullptr);
        }
        return u;
    };
    auto store_stage = [&](const RSet& q, unsigned char* st) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < CPS; ++u) {
            unsigned char* As = st + u * SUB;
            unsigned char* Bs = As + BM * RS;
            *(uint4*)(As + r0 * RS + chunk * 16) = act_u4(q.a0[u]);
            *(uint4*)(As + (r0 + 64) * RS + chunk * 16) = act_u4(q.a1[u]);
            if (b_thread) {
                *(uint4*)(Bs + r0 * RS + chunk * 16) = q.b0[u];
                if (BROWS > 1) *(uint4*)(Bs + (r0 + 64) * RS + chunk * 16) = q.b1[u];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline: double-buffered LDS, one barrier per stage, prefetch distance 2 (the loads of stage s+2 are
    // issued before the MFMAs of stage s and land in LDS after the MFMAs of stage s+1).
    // split-K (blockIdx.y): this block walks stages [sbase, sbase + nstages) of the K loop
    const int all_stages = (total + CPS - 1) / CPS;
    const int sbase = p.splits > 1 ? blockIdx.y * p.stages_per_split : 0;
    const int nstages = p.splits > 1 ? min(p.stages_per_split, all_stages - sbase) : all_stages;
    load_stage(s0, sbase);
    store_stage(s0, smem);
    if (nstages > 1) load_stage(s1, sbase + 1);
    __syncthreads();
    auto iter = [&](int sidx, RSet& ld_set, const RSet& st_set) __attribute__((always_inline)) {
        if (sidx + 2 < nstages) load_stage(ld_set, sbase + sidx + 2);
        const unsigned char* st = smem + (sidx & 1) * STAGE;
#pragma unroll
        for (int u = 0; u < CPS; ++u)
            Mma<T>::template chunk<TM, TN>(st + u * SUB + (wm * TM * 32) * RS, st + u * SUB + BM * RS + (wn * TN * 32) * RS,
                                           lane, acc);
        if (sidx + 1 < nstages) store_stage(st_set, smem + ((sidx + 1) & 1) * STAGE);
        __syncthreads();
    };
    for (int sidx = 0; sidx < nstages; sidx += 2) {
        iter(sidx, s0, s1);
        if (sidx + 1 < nstages) iter(sidx + 1, s1, s0);
    }

    if (p.splits > 1) {
        // split-K partial: raw fp32 accumulators, [split][m][ldw]; bias / CoordConv / act' / residual in the reduce kernel
        float* wsp = p.ws + ((long long)blockIdx.y * M) * p.ldw;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = nt * BN + (wn * TN + tn) * 32 + (lane & 31);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mt * BM + (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (m < M) wsp[(long long)m * p.ldw + col] = acc[tm][tn][e];
                }
        }
        return;
    }
    // ---- epilogue: row table in LDS (pixel index, CoordConv class, j, i)
    if (tid < BM) {
        const int m = mt * BM + tid;
        long long pix = -1;
        int cls = 0, i = 0, j = 0;
        if (m < M) {
            const int img = m / hw_o, rem = m - img * hw_o;
            i = rem / p.wo; j = rem - i * p.wo;
            pix = ((long long)img * p.out_h + (i * p.out_sy + p.out_oy)) * p.out_w + (j * p.out_sx + p.out_ox);
            if (p.coord_tab) {
                int ym = 0, xm = 0;
                for (int r = 0; r < p.kh; ++r) ym |= ((unsigned)(i * p.in_sy + tap_dy_of(p.tap_off, r * p.kw)) < (unsigned)p.hi) << r;
                for (int s = 0; s < p.kw; ++s) xm |= ((unsigned)(j * p.in_sx + tap_dx_of(p.tap_off, s)) < (unsigned)p.wi) << s;
                cls = ym * 8 + xm;
            }
        }
        rowpix[tid] = pix;
        rowaux[tid * 3 + 0] = cls; rowaux[tid * 3 + 1] = j; rowaux[tid * 3 + 2] = i;
    }
    __syncthreads();

    T* __restrict__ outT = (T*)p.out;
    float* __restrict__ outF = (float*)p.out;
    const T* __restrict__ res = (const T*)p.res;
    const T* __restrict__ dact = (const T*)p.dact;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = nt * BN + (wn * TN + tn) * 32 + (lane & 31);
        const bool cvalid = col < p.co;
        const bool cfill = col < p.co_fill;
        const float bias = (cvalid && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const long long pix = rowpix[rl];
                if (pix < 0 || !cfill) continue;
                float v = 0.f;
                if (cvalid) {
                    v = acc[tm][tn][e] + bias;
                    if (p.coord_tab) {
                        const float* tb = p.coord_tab + (long long)rowaux[rl * 3] * 3 * p.co + col;
                        v += tb[0] + (float)rowaux[rl * 3 + 1] * tb[p.co] + (float)rowaux[rl * 3 + 2] * tb[2 * p.co];
                    }
                    if (dact) v *= (ld_as_float<T>(dact + pix * p.ldd + col) > 0.f) ? 1.f : dact_ns;
                    if (res) {
                        float rr = ld_as_float<T>(res + pix * p.ldr + col);
                        if (p.res_act) rr = rr > 0.f ? rr : rr / p.act_slope;      // residual stored as leaky-ReLU(x)
                        v += rr;
                    }
                    if (p.out_act) v = ups_vmax(v, ups_slope_eff(p.out_act, p.act_slope) * v);   // post-activation storage
                }
                if (p.out_f32) outF[pix * p.ldo + col] = v;
                else st_from_float<T>(outT + pix * p.ldo + col, v);
            }
        }
    }
}

// Deterministic reduce of the split-K partials + the full epilogue (one thread per output element, columns fastest).
template <typename T>
__global__ __launch_bounds__(256) void igemm_splitk_epilogue(const ConvK p, const int M) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int col = (int)(gid % p.co_fill);
    const long long m = gid / p.co_fill;
    if (m >= M) return;
    float v = 0.f;
    for (int sp = 0; sp < p.splits; ++sp) v += p.ws[((long long)sp * M + m) * p.ldw + col];
    const int hw_o = p.ho * p.wo;
    const int img = (int)(m / hw_o), rem = (int)(m - (long long)img * hw_o);
    const int i = rem / p.wo, j = rem - i * p.wo;
    const long long pix = ((long long)img * p.out_h + (i * p.out_sy + p.out_oy)) * p.out_w + (j * p.out_sx + p.out_ox);
    const bool cvalid = col < p.co;
    if (cvalid) {
        if (p.bias) v += p.bias[col];
        if (p.coord_tab) {
            int ym = 0, xm = 0;
            for (int r = 0; r < p.kh; ++r) ym |= ((unsigned)(i * p.in_sy + tap_dy_of(p.tap_off, r * p.kw)) < (unsigned)p.hi) << r;
            for (int q = 0; q < p.kw; ++q) xm |= ((unsigned)(j * p.in_sx + tap_dx_of(p.tap_off, q)) < (unsigned)p.wi) << q;
            const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col;
            v += tb[0] + (float)j * tb[p.co] + (float)i * tb[2 * p.co];
        }
        if (p.dact) v *= (ld_as_float<T>((const T*)p.dact + pix * p.ldd + col) > 0.f) ? 1.f : ups_slope_eff(p.dact_kind, p.act_slope);
        if (p.res) {
            float rr = ld_as_float<T>((const T*)p.res + pix * p.ldr + col);
            if (p.res_act) rr = rr > 0.f ? rr : rr / p.act_slope;
            v += rr;
        }
        if (p.out_act) v = ups_vmax(v, ups_slope_eff(p.out_act, p.act_slope) * v);
    } else {
        v = 0.f;
    }
    if (p.out_f32) ((float*)p.out)[pix * p.ldo + col] = v;
    else st_from_float<T>((T*)p.out + pix * p.ldo + col, v);
}

template <typename T>
int launch(const ups_conv_desc& dd, hipStream_t s) {
    constexpr int EPC = Chunk<T>::N;
    ConvK d;
    d.n = dd.n; d.hi = dd.hi; d.wi = dd.wi; d.ci = dd.ci; d.ldi = dd.ldi; d.ho = dd.ho; d.wo = dd.wo; d.co = dd.co;
    d.co_fill = dd.co_fill; d.ldo = dd.ldo; d.out_h = dd.out_h; d.out_w = dd.out_w; d.out_sy = dd.out_sy;
    d.out_sx = dd.out_sx; d.out_oy = dd.out_oy; d.out_ox = dd.out_ox; d.in_sy = dd.in_sy; d.in_sx = dd.in_sx;
    d.ntaps = dd.ntaps; d.kh = dd.kh; d.kw = dd.kw; d.act_in = dd.act_in; d.out_f32 = dd.out_f32;
    d.dact_kind = dd.dact_kind; d.ldr = dd.ldr; d.ldd = dd.ldd; d.act_slope = dd.act_slope;
    d.out_act = dd.out_act; d.res_act = dd.res_act;
    d.in = dd.in; d.w = dd.w; d.out = dd.out; d.bias = dd.bias; d.coord_tab = dd.coord_tab; d.res = dd.res; d.dact = dd.dact;
    d.tap_off = 0; d.tap_wi = 0;
    for (int t = 0; t < dd.ntaps; ++t) {
        if (dd.tap_dy[t] < -1 || dd.tap_dy[t] > 2 || dd.tap_dx[t] < -1 || dd.tap_dx[t] > 2 || dd.tap_w[t] < 0 || dd.tap_w[t] > 15)
            return UPS_E_ARG;
        d.tap_off |= (unsigned long long)(((dd.tap_dy[t] + 1) << 2) | (dd.tap_dx[t] + 1)) << (4 * t);
        d.tap_wi |= (unsigned long long)dd.tap_w[t] << (4 * t);
    }
    const long long M = (long long)d.n * d.ho * d.wo;
    if (M <= 0 || M > 0x7fffffffLL) return UPS_E_ARG;
    const int kchunks = ups_cdiv(d.ci, 4 * EPC);
    const int ctot = d.co_fill;
    const int mtiles = ups_cdiv(M, BM);
    // tile width: 128 when the grid fills the chip, narrower tiles (more blocks) for skinny problems; skinny problems
    // with long K loops also stage several chunks per barrier (CPS)
    int bn = ctot > 64 ? 128 : (ctot > 32 ? 64 : 32);
    if (bn == 128 && mtiles * ups_cdiv(ctot, 128) < 256) bn = 64;
    if (bn == 64 && mtiles * ups_cdiv(ctot, 64) < 256) bn = 32;
    {   // UPS_IGEMM_FORCE="<bn>,<cps>": tuning override (one of 128,1 / 64,2 / 64,1 / 32,4 / 32,1)
        static int fbn = -1, fcps = -1;
        if (fbn < 0) {
            const char* e = getenv("UPS_IGEMM_FORCE");
            fbn = 0;
            if (e) sscanf(e, "%d,%d", &fbn, &fcps);
        }
        if (fbn > 0) bn = fbn;
    }
    const int ntn = ups_cdiv(ctot, bn);
    const bool skinny = mtiles * ntn < 1024 && d.ntaps * kchunks >= 8;
    int cps = !skinny ? 1 : (bn == 32 ? 4 : (bn == 64 ? 2 : 1));
    {
        const char* e = getenv("UPS_IGEMM_FORCE");
        int a = 0, b = 0;
        if (e && sscanf(e, "%d,%d", &a, &b) == 2 && b > 0) cps = b;
    }
    // split-K for latency-bound problems: few tiles, long K loop, workspace supplied.  Wider tiles + more blocks.
    d.ws = nullptr; d.splits = 1; d.stages_per_split = 0; d.ldw = 0;
    {
        static int sk_on = -1;
        if (sk_on < 0) { const char* e = getenv("UPS_NO_SPLITK"); sk_on = (e && e[0] == '1') ? 0 : 1; }
        const int chunks = d.ntaps * kchunks;
        const int bn_sk = ctot > 32 ? 64 : 32;
        const int blocks_sk = mtiles * ups_cdiv(ctot, bn_sk);
        // (a) fewer than 192 tiles: split up to ~512 blocks; (b) long K loops on grids that fill the chip at most about
        // twice: split up to ~1024 blocks (measured: 4x4 / 8x8 layers of the encoders, dv and the VGG trunk)
        const bool few = blocks_sk < 192 && chunks >= 16;
        const bool longk = (blocks_sk <= 320 && chunks >= 64) || (blocks_sk <= 512 && chunks >= 128);
        if (sk_on && dd.workspace && (few || longk)) {
            const int cps_sk = bn_sk == 64 ? 2 : 4;
            const int stages = ups_cdiv(chunks, cps_sk);
            int splits = ups_cdiv(few ? 512 : 1024, blocks_sk);
            if (splits > stages / 2) splits = stages / 2;
            const int sps = ups_cdiv(stages, splits);
            splits = ups_cdiv(stages, sps);
            const int ldw = ups_cdiv(ctot, bn_sk) * bn_sk;
            const size_t need = (size_t)splits * (size_t)M * ldw * sizeof(float);
            if (splits >= 2 && need <= dd.workspace_bytes) {
                bn = bn_sk; d.ws = dd.workspace; d.splits = splits; d.stages_per_split = sps; d.ldw = ldw;
            }
        }
    }
    const int ntn_l = ups_cdiv(ctot, bn);
    const int cps_l = d.splits > 1 ? (bn == 64 ? 2 : 4) : cps;
#define UPS_LAUNCH_IG(BNV, CPSV)                                                                                      \
    do {                                                                                                              \
        const size_t shmem = 2 * (size_t)(CPSV) * (BM + (BNV)) * RS + BM * 20;                                        \
        static UpsPerDevice attr_done;                                                                                \
        if (!attr_done) {                                                                                             \
            if (hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BNV, CPSV>,                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)           \
                return UPS_E_LAUNCH;                                                                                  \
            attr_done = true;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL((conv_igemm_kernel<T, BNV, CPSV>), dim3(mtiles * ntn_l, d.splits), dim3(256), shmem, s, d,  \
                           (int)M, ntn_l, kchunks);                                                                   \
    } while (0)
    if (bn == 128) UPS_LAUNCH_IG(128, 1);
    else if (bn == 64 && cps_l == 2) UPS_LAUNCH_IG(64, 2);
    else if (bn == 64) UPS_LAUNCH_IG(64, 1);
    else if (cps_l == 4) UPS_LAUNCH_IG(32, 4);
    else UPS_LAUNCH_IG(32, 1);
#undef UPS_LAUNCH_IG
    if (d.splits > 1) {
        const long long elems = M * (long long)d.co_fill;
        hipLaunchKernelGGL((igemm_splitk_epilogue<T>), dim3(ups_cdiv(elems, 256)), dim3(256), 0, s, d, (int)M);
    }
    return UPS_OK;
}

}  // namespace

int ups_conv3x3_patch_try(const ups_conv_desc* d, hipStream_t s);   // conv3x3_patch.hip
bool ups_conv3x3_patch_signs(const ups_conv_desc* d);               // conv3x3_patch.hip
extern "C" int ups_sign_pack(const void* x, int32_t dtype, int64_t chunks, void* sign_bits, void* stream);   // pointwise.hip

// ups_conv_desc.sign_out is written by the kernels that have the output tile in hand (the 16-bit patch kernel); a launch that went to
// another kernel leaves it untouched -- a pass over the finished output would cost the read the bits are there to save -- and says
// so: ups_conv_sign_out_written() reports on the calling thread's last ups_conv_igemm call.
thread_local int g_ups_sign_written = 0;        // (set to 1 by the launcher of a kernel that writes the bits)
extern "C" int ups_conv_sign_out_written(void) { return g_ups_sign_written; }
static int sign_out_pass(const ups_conv_desc* d, void* stream) { (void)d; (void)stream; return UPS_OK; }
int ups_conv3x3_first_try(const ups_conv_desc* d, hipStream_t s);   // conv3x3_first.hip
int ups_conv3x3_s2_try(const ups_conv_desc* d, hipStream_t s);      // conv3x3_s2.hip
int ups_conv3x3_rows_try(const ups_conv_desc* d, hipStream_t s);    // conv3x3_rows.hip
int ups_conv3x3_rows_s2_try(const ups_conv_desc* d, hipStream_t s);
int ups_conv3x3_thinout_try(const ups_conv_desc* d, hipStream_t s);
int ups_conv3x3_rows_maskgrad_try(const ups_conv_desc* d, hipStream_t s);

extern "C" int ups_conv_igemm(const ups_conv_desc* d, void* stream) {
    UPS_CHECK_ARG(d != nullptr);
    g_ups_sign_written = 0;
    UPS_CHECK_ARG(d->dtype == UPS_F32 || d->dtype == UPS_BF16 || d->dtype == UPS_F16);
    UPS_CHECK_ARG(d->in && d->w && (d->out || d->mask_grad));
    if (d->dtype == UPS_F16 && (d->d2s || d->f8_deq || d->mask_bits || d->mask_grad || d->in_f8 || d->out_f8 || d->out_f8_amax)) {
        ups_set_error("ups_conv_igemm: UPS_F16 is a forward format: no fp8 copies, part masks or depth-to-space output");
        return UPS_E_UNSUPPORTED;
    }
    UPS_CHECK_ARG(d->ci > 0 && d->ci % 8 == 0 && d->ldi % 8 == 0 && d->ci <= d->ldi);
    UPS_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= 9);
    UPS_CHECK_ARG(d->act_slope >= 0.f && d->act_slope <= 1.f);    // activation-on-load is max(x, slope * x)
    UPS_CHECK_ARG(d->out_act >= UPS_ACT_NONE && d->out_act <= UPS_ACT_RELU);
    UPS_CHECK_ARG(d->res_act == UPS_ACT_NONE || (d->res_act == UPS_ACT_LRELU && d->act_slope > 0.f && d->res));   // invertible only
    UPS_CHECK_ARG(d->co >= 1 && d->co_fill >= d->co && (d->d2s ? d->d2s <= d->ldo : d->co_fill <= d->ldo));
    UPS_CHECK_ARG(d->n > 0 && d->ho > 0 && d->wo > 0 && d->hi > 0 && d->wi > 0);
    UPS_CHECK_ARG(!d->coord_tab || (d->kh * d->kw == d->ntaps && d->kh <= 3 && d->kw <= 3));
    UPS_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->w & 15) == 0);
    UPS_CHECK_ARG((d->out_sy * (d->ho - 1) + d->out_oy) < d->out_h && (d->out_sx * (d->wo - 1) + d->out_ox) < d->out_w);
    if (d->sign_out) {      // sign bits of a plain 16-bit output tensor
        UPS_CHECK_ARG(d->dtype != UPS_F32 && !d->out_f32 && (d->ldo & 7) == 0 && d->out && !d->d2s && !d->mask_grad);
        UPS_CHECK_ARG(d->out_sy == 1 && d->out_sx == 1 && !d->out_oy && !d->out_ox && d->out_h == d->ho && d->out_w == d->wo);
    }
    // 3x3 / stride-1 problems on 16-aligned images go to the patch-tiled kernel (halo reuse across the 9 taps)
    const char* force = getenv("UPS_FORCE_GENERIC_CONV");
    if (!(force && force[0] == '1')) {
        // first layers (<= 8 input channels, 32 / 64 outputs): the im2col-in-the-fragment kernel, an output-write stream
        if (ups_conv3x3_first_try(d, (hipStream_t)stream) == 0) { UPS_LAUNCH_CHECK(); return UPS_OK; }
        // the large 3x3 / stride-2 `downsample` forwards (32 / 64 input channels): taps straight from global memory, no gather
        {
            const int rr = ups_conv3x3_rows_s2_try(d, (hipStream_t)stream);      // (the two encoder shapes as row streams)
            if (rr == 0) { UPS_LAUNCH_CHECK(); return UPS_OK; }
            if (rr < 0) { ups_set_error("ups_conv_igemm: row-streaming stride-2 kernel launch setup failed"); return rr; }
        }
        {
            const int sr = ups_conv3x3_s2_try(d, (hipStream_t)stream);
            if (sr == 0) { UPS_LAUNCH_CHECK(); return sign_out_pass(d, stream); }
            if (sr < 0) { ups_set_error("ups_conv_igemm: stride-2 kernel launch setup failed"); return sr; }
        }
        // thin residual blocks on large batches of full-width rows: the row-streaming kernel; the K-deep logit convolution: its
        // strip form
        {
            int rr = ups_conv3x3_thinout_try(d, (hipStream_t)stream);
            if (rr == 1) rr = ups_conv3x3_rows_try(d, (hipStream_t)stream);
            if (rr == 1) rr = ups_conv3x3_rows_maskgrad_try(d, (hipStream_t)stream);
            if (rr == 0) { UPS_LAUNCH_CHECK(); return UPS_OK; }
            if (rr < 0) { ups_set_error("ups_conv_igemm: row-streaming kernel launch setup failed"); return rr; }
        }
        const int pr = ups_conv3x3_patch_try(d, (hipStream_t)stream);
        if (pr == 0) { UPS_LAUNCH_CHECK(); if (d->sign_out && ups_conv3x3_patch_signs(d)) g_ups_sign_written = 1; return UPS_OK; }
        if (pr < 0) { ups_set_error("ups_conv_igemm: patch kernel launch setup failed"); return pr; }
    }
    if (d->d2s) {
        ups_set_error("ups_conv_igemm: the depth-to-space output needs the bf16 3x3 / stride-1 patch kernel (16-aligned lattice, 4 x 2^k channels)");
        return UPS_E_UNSUPPORTED;
    }
    if (d->f8_deq) {
        ups_set_error("ups_conv_igemm: the fp8 forward needs the bf16 3x3 / stride-1 patch kernel (16-aligned images, ci %% 64 == 0, no mask)");
        return UPS_E_UNSUPPORTED;
    }
    if (d->mask_bits || d->mask_grad) {
        ups_set_error("ups_conv_igemm: the part-masked forms need the bf16 3x3 / stride-1 patch kernel (16-aligned images, P <= 32)");
        return UPS_E_UNSUPPORTED;
    }
    if (d->out_f8 || d->out_f8_amax || d->in_f8) {   // (the generic kernel never writes / reads fp8 copies)
        ups_set_error("ups_conv_igemm: fp8 copies (in_f8 / out_f8 / out_f8_amax) need the bf16 3x3 / stride-1 patch kernel");
        return UPS_E_UNSUPPORTED;
    }
    int rc = (d->dtype == UPS_F32) ? launch<float>(*d, (hipStream_t)stream)
             : (d->dtype == UPS_F16 ? launch<f16>(*d, (hipStream_t)stream) : launch<bf16>(*d, (hipStream_t)stream));
    if (rc != UPS_OK) { ups_set_error("ups_conv_igemm: bad problem size"); return rc; }
    UPS_LAUNCH_CHECK();
    return sign_out_pass(d, stream);
}
