// Helpers around the convolution engine: weight layout/dtype preparation, the CoordConv affine table and its
// gradient, bias gradients.  (cub/code/nn.py:617-664 conv variables V [kh,kw,Cin(+2),Cout] + b; CoordConv
// channels nn.py:2123-2154.)
#include "common.h"

namespace {

// Blocked-K weight layout shared by every convolution kernel: [tap][k-chunk][row][64 bytes], so that the weight tile
// of one (tap, k-chunk) -- rows x 64 B -- is one contiguous, fully coalesced range (whole cache lines, all L2 channels).
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ src, int ntaps, int cin_v, int ci_log, int co,
                                   T* __restrict__ wf, int ci_pad, T* __restrict__ wd, int drows, int dk) {
    constexpr int BK = 64 / (int)sizeof(T);
    const int kcf = (ci_pad + BK - 1) / BK, kcd = (dk + BK - 1) / BK;
    const long long nf = wf ? (long long)ntaps * kcf * co * BK : 0;
    const long long nd = wd ? (long long)ntaps * kcd * drows * BK : 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd;
         idx += (long long)gridDim.x * blockDim.x) {
        if (idx < nf) {
            const int kk = (int)(idx % BK);
            long long t = idx / BK;
            const int c = (int)(t % co); t /= co;
            const int kc = (int)(t % kcf), tap = (int)(t / kcf);
            const int k = kc * BK + kk;
            const float v = k < ci_log ? src[((long long)tap * cin_v + k) * co + c] : 0.f;
            st_from_float<T>(wf + idx, v);
        } else {
            const long long j = idx - nf;
            const int kk = (int)(j % BK);
            long long t = j / BK;
            const int r = (int)(t % drows); t /= drows;
            const int kc = (int)(t % kcd), tap = (int)(t / kcd);
            const int k = kc * BK + kk;
            const float v = (k < co && r < ci_log) ? src[((long long)tap * cin_v + r) * co + k] : 0.f;
            st_from_float<T>(wd + j, v);
        }
    }
}

__host__ __device__ inline long long prep_elems(const ups_prep_item& it, int bk) {
    const long long nf = it.w_fwd ? (long long)it.ntaps * ((it.ci_pad + bk - 1) / bk) * it.co * bk : 0;
    const long long nd = it.w_dgrad ? (long long)it.ntaps * ((it.dgrad_k + bk - 1) / bk) * it.dgrad_rows * bk : 0;
    const long long nc = it.ctab ? (long long)64 * it.co : 0;
    return nf + nd + nc;
}

// One CoordConv table entry (class = which taps are inside the image, channel c): the affine form k0 + kj * j + ki * i of the two
// coordinate channels' contribution.  ONE definition with explicit fused operations for the per-layer launch and the batched one:
// left to the compiler's contraction the two call sites rounded differently, and a model restored from a checkpoint (tables from
// the per-layer launch) was an ulp away from the run that wrote it (tables from the batched launch after its last step).
__device__ __forceinline__ void coord_table_entry(const float* __restrict__ V, int kh, int kw, int cin_v, int ci_log, int co, int c,
                                                  int ym, int xm, const int* dy, const int* dx, int in_sy, int in_sx, float ax, float ay,
                                                  float& k0, float& kj, float& ki) {
    k0 = 0.f; kj = 0.f; ki = 0.f;
    const float sxj = __fmul_rn(ax, (float)in_sx), syi = __fmul_rn(ay, (float)in_sy);
    for (int r = 0; r < kh; ++r) {
        if (!((ym >> r) & 1)) continue;
        for (int s = 0; s < kw; ++s) {
            if (!((xm >> s) & 1)) continue;
            const float vx = V[((long long)(r * kw + s) * cin_v + ci_log) * co + c];
            const float vy = V[((long long)(r * kw + s) * cin_v + ci_log + 1) * co + c];
            const int dys = r == 0 ? dy[0] : (r == 1 ? dy[1] : dy[2]);
            const int dxs = s == 0 ? dx[0] : (s == 1 ? dx[1] : dx[2]);
            k0 = __fmaf_rn(__fmaf_rn(ax, (float)dxs, -1.f), vx, k0);
            k0 = __fmaf_rn(__fmaf_rn(ay, (float)dys, -1.f), vy, k0);
            kj = __fmaf_rn(sxj, vx, kj);
            ki = __fmaf_rn(syi, vy, ki);
        }
    }
}

// one launch for all layers: a 256-element chunk -> its item by binary search in the chunk prefix.  (Round 4: every block of the
// first form searched the prefix in global memory -- eight dependent loads, ~4 us, before it converted its 256 elements: 243 447
// blocks, 0.31 ms for 0.27 GB.  Now the prefix sits in LDS and a block converts CPB consecutive chunks.)
template <typename T>
__device__ __forceinline__ void weight_prep_chunk(const ups_prep_item& it, long long blk0, float (*tile)[9]) {
    constexpr int BK = 64 / (int)sizeof(T);
    const long long idx = blk0 + threadIdx.x;
    const int kcf = (it.ci_pad + BK - 1) / BK, kcd = (it.dgrad_k + BK - 1) / BK;
    const long long nf = it.w_fwd ? (long long)it.ntaps * kcf * it.co * BK : 0;
    const long long nd = it.w_dgrad ? (long long)it.ntaps * kcd * it.dgrad_rows * BK : 0;
    const long long nc = it.ctab ? (long long)64 * it.co : 0;
    // forward layout [tap][k-chunk][co][BK]: the 256 outputs of a chunk are 8 output channels x 32 k of one (tap, k-chunk) -- a
    // 32 x 8 patch of the [k][co] source: read along co, turned in LDS, written along k (16-bit layouts, co % 8 == 0)
    if (sizeof(T) == 2 && (it.co & 7) == 0 && blk0 + 256 <= nf && nf < (1ll << 31)) {
        unsigned t = (unsigned)(blk0 / BK);
        const int c0 = (int)(t % (unsigned)it.co); t /= (unsigned)it.co;
        const int kc = (int)(t % (unsigned)kcf), tap = (int)(t / (unsigned)kcf);
        const int rk = threadIdx.x >> 3, rc = threadIdx.x & 7;
        const int k = kc * BK + rk;
        __syncthreads();                       // (the previous chunk's reads of the tile)
        tile[rk][rc] = k < it.ci_log ? it.src[((long long)tap * it.cin_v + k) * it.co + c0 + rc] : 0.f;
        __syncthreads();
        st_from_float<T>((T*)it.w_fwd + idx, tile[threadIdx.x & 31][threadIdx.x >> 5]);
        return;
    }
    // (32-bit index arithmetic where a layer's copies stay below 2^31 elements -- every layer of the shipped configs: 64-bit
    // divisions by run-time values cost ~100 instructions each)
    const bool small = nf + nd + nc < (1ll << 31);
    if (idx < nf) {
        int kk, c, kc, tap;
        if (small) {
            const unsigned u = (unsigned)idx;
            kk = (int)(u % BK);
            unsigned t = u / BK;
            c = (int)(t % (unsigned)it.co); t /= (unsigned)it.co;
            kc = (int)(t % (unsigned)kcf); tap = (int)(t / (unsigned)kcf);
        } else {
            kk = (int)(idx % BK);
            long long t = idx / BK;
            c = (int)(t % it.co); t /= it.co;
            kc = (int)(t % kcf); tap = (int)(t / kcf);
        }
        const int k = kc * BK + kk;
        st_from_float<T>((T*)it.w_fwd + idx, k < it.ci_log ? it.src[((long long)tap * it.cin_v + k) * it.co + c] : 0.f);
    } else if (idx < nf + nd) {
        const long long j = idx - nf;
        int kk, r, kc, tap;
        if (small) {
            const unsigned u = (unsigned)j;
            kk = (int)(u % BK);
            unsigned t = u / BK;
            r = (int)(t % (unsigned)it.dgrad_rows); t /= (unsigned)it.dgrad_rows;
            kc = (int)(t % (unsigned)kcd); tap = (int)(t / (unsigned)kcd);
        } else {
            kk = (int)(j % BK);
            long long t = j / BK;
            r = (int)(t % it.dgrad_rows); t /= it.dgrad_rows;
            kc = (int)(t % kcd); tap = (int)(t / kcd);
        }
        const int k = kc * BK + kk;
        st_from_float<T>((T*)it.w_dgrad + j, (k < it.co && r < it.ci_log) ? it.src[((long long)tap * it.cin_v + r) * it.co + k] : 0.f);
    } else if (idx < nf + nd + nc) {
        const long long j = idx - nf - nd;
        const int c = (int)(j % it.co), cls = (int)(j / it.co);
        const int ym = cls >> 3, xm = cls & 7;
        float k0, kj, ki;
        coord_table_entry(it.src, it.kh, it.kw, it.cin_v, it.ci_log, it.co, c, ym, xm, it.dy, it.dx, it.in_sy, it.in_sx, it.ax, it.ay,
                          k0, kj, ki);
        it.ctab[((long long)cls * 3 + 0) * it.co + c] = k0;
        it.ctab[((long long)cls * 3 + 1) * it.co + c] = kj;
        it.ctab[((long long)cls * 3 + 2) * it.co + c] = ki;
    }
}

constexpr int PREP_CPB = 16;        // chunks per block
constexpr int PREP_LDS_ITEMS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const ups_prep_item* __restrict__ items,
                                                                const long long* __restrict__ prefix, int n_items, long long total_chunks) {
    __shared__ long long spre[PREP_LDS_ITEMS];
    __shared__ float tile[32][9];
    __shared__ float tile4[32][33];
    const bool in_lds = n_items <= PREP_LDS_ITEMS;
    if (in_lds) {
        for (int i = threadIdx.x; i < n_items; i += 256) spre[i] = prefix[i];
        __syncthreads();
    }
    const long long* pre = in_lds ? (const long long*)spre : prefix;
    int lo = -1;
    ups_prep_item it;
    for (int j = 0; j < PREP_CPB; ++j) {
        const long long b = (long long)blockIdx.x * PREP_CPB + j;
        if (b >= total_chunks) break;
        if (lo < 0 || (lo + 1 < n_items && pre[lo + 1] <= b)) {
            int l = 0, h = n_items - 1;
            while (l < h) {
                const int mid = (l + h + 1) >> 1;
                if (pre[mid] <= b) l = mid; else h = mid - 1;
            }
            lo = l;
            it = items[lo];
        }
        // four aligned chunks of the forward layout = 32 output channels x 32 k of one (tap, k-chunk): read as 128-byte rows of the
        // [k][co] source (the 8-channel form reads 32-byte pieces -- a quarter of a line each, of the 33152-wide head's rows above all)
        constexpr int BK = 64 / (int)sizeof(T);
        const long long c0i = b - pre[lo];
        if (sizeof(T) == 2 && (it.co & 31) == 0 && (c0i & 3) == 0 && j + 4 <= PREP_CPB && b + 4 <= total_chunks && it.w_fwd &&
            (((unsigned long long)it.src) & 15ull) == 0) {
            const int kcf = (it.ci_pad + BK - 1) / BK;
            const long long nf = (long long)it.ntaps * kcf * it.co * BK;
            if ((c0i + 4) * 256 <= nf && nf < (1ll << 31)) {
                unsigned t = (unsigned)(c0i * 256 / BK);
                const int c0 = (int)(t % (unsigned)it.co); t /= (unsigned)it.co;
                const int kc = (int)(t % (unsigned)kcf), tap = (int)(t / (unsigned)kcf);
                const int rk = threadIdx.x >> 3, rc = (threadIdx.x & 7) * 4;
                const int k = kc * BK + rk;
                __syncthreads();
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < it.ci_log) v = *(const float4*)(it.src + ((long long)tap * it.cin_v + k) * it.co + c0 + rc);
                tile4[rk][rc] = v.x; tile4[rk][rc + 1] = v.y; tile4[rk][rc + 2] = v.z; tile4[rk][rc + 3] = v.w;
                __syncthreads();
                T* dst = (T*)it.w_fwd + c0i * 256;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = q * 256 + threadIdx.x;         // element (c = e / 32, kk = e % 32) of the 32 x 32 unit
                    st_from_float<T>(dst + e, tile4[e & 31][e >> 5]);
                }
                j += 3;
                continue;
            }
        }
        weight_prep_chunk<T>(it, c0i * 256, tile);
    }
}

struct Taps3 { int dy[3], dx[3]; };

__global__ void coord_table_kernel(const float* __restrict__ V, int kh, int kw, int ci_log, int co, Taps3 tp, int in_sy,
                                   int in_sx, float ax, float ay, float* __restrict__ tab) {
    const int cls = blockIdx.y;
    const int ym = cls >> 3, xm = cls & 7;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= co) return;
    float k0, kj, ki;
    coord_table_entry(V, kh, kw, ci_log + 2, ci_log, co, c, ym, xm, tp.dy, tp.dx, in_sy, in_sx, ax, ay, k0, kj, ki);
    tab[((long long)cls * 3 + 0) * co + c] = k0;
    tab[((long long)cls * 3 + 1) * co + c] = kj;
    tab[((long long)cls * 3 + 2) * co + c] = ki;
}

// gsum[pix][c] = sum_n d[n][pix][c]: one thread per (pixel, 16-byte channel chunk), n independent 16-B loads in flight
template <typename T>
__global__ void batch_sum_kernel(const T* __restrict__ d, int n, long long pix, int co, int ldo, float* __restrict__ g) {
    constexpr int E = Chunk<T>::N;
    const int cpr = (co + E - 1) / E;
    const long long total = pix * cpr;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % cpr);
        const long long px = idx / cpr;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const T* base = d + px * ldo + k * E;
        // (round 6: `unroll 16` measured SLOWER -- 47.9 against 25.7 us per launch on average over the step's 27 launches)
#pragma unroll 4
        for (int b = 0; b < n; ++b) {
            float f[E];
            const uint4 u = *(const uint4*)(base + (long long)b * pix * ldo);
            Chunk<T>::unpack(u, f);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (k * E + e < co) g[px * co + k * E + e] = acc[e];
    }
}

// CoordConv weight gradient, separable two-stage reduction over the batch-summed gradient map g[ho][wo][co]:
//   stage 1 (rows):  A[r][j][c] = sum_i valid_y(i,r) g[i][j][c],  B[r][j][c] = sum_i valid_y(i,r) (ay*y-1) g[i][j][c]
//                    (r = kh: all rows, used for the bias)
//   stage 2 (cols):  dVx[r][s][c] = sum_j valid_x(j,s) (ax*x-1) A[r][j][c],  dVy[r][s][c] = sum_j valid_x(j,s) B[r][j][c]
// (round 4: both kernels ran one thread per output with a serial loop over the 128 rows / columns -- 16 blocks, 48 + 36 us per layer at
// 128x128, and on the weight-gradient queue's tail at the end of the step nothing hides them.  Now 8 / 4 row groups per output,
// combined through LDS in a fixed order: deterministic, 128 blocks, loads of a group independent of each other.)
__global__ __launch_bounds__(256) void coord_rows_kernel(const float* __restrict__ g, int hi, int ho, int wo, int co, int kh, Taps3 tp,
                                                         int in_sy, float ay, float* __restrict__ scratch) {
    __shared__ float red[8][7][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long long plane = (long long)wo * co;
    const long long idx = (long long)blockIdx.x * 32 + o;
    const bool ok = idx < plane;
    float A[4] = {0.f, 0.f, 0.f, 0.f}, B[3] = {0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll 4
        for (int i = grp; i < ho; i += 8) {
            const float v = g[(long long)i * plane + idx];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r < kh) {
                    const int y = i * in_sy + (r == 0 ? tp.dy[0] : (r == 1 ? tp.dy[1] : tp.dy[2]));
                    if ((unsigned)y < (unsigned)hi) { A[r] += v; B[r] += (ay * (float)y - 1.f) * v; }
                }
            }
            A[3] += v;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { red[grp][2 * r][o] = A[r]; red[grp][2 * r + 1][o] = B[r]; }
    red[grp][6][o] = A[3];
    __syncthreads();
    if (grp == 0 && ok) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            float t = red[0][q][o];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[k][q][o];
            if (q < 6) { if ((q >> 1) < kh) scratch[(long long)q * plane + idx] = t; }
            else scratch[(2 * kh) * plane + idx] = t;
        }
    }
}

__global__ __launch_bounds__(256) void coord_cols_kernel(const float* __restrict__ scratch, int wi, int wo, int co, int kh, int kw, Taps3 tp,
                                                         int in_sx, float ax, int ci_log, float* __restrict__ gV, float* __restrict__ gb) {
    __shared__ float red[4][2][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int tap = blockIdx.y;                  // kh*kw taps, then one extra block row for the bias
    const long long plane = (long long)wo * co;
    const bool ok = c < co;
    float sx = 0.f, sy = 0.f;
    if (tap == kh * kw) {
        if (!gb) return;                         // (uniform)
        if (ok) {
#pragma unroll 4
            for (int j = grp; j < wo; j += 4) sx += scratch[(2 * kh) * plane + (long long)j * co + c];
        }
    } else if (ok) {
        const int r = tap / kw, s_ = tap % kw;
        const int dxs = s_ == 0 ? tp.dx[0] : (s_ == 1 ? tp.dx[1] : tp.dx[2]);
#pragma unroll 4
        for (int j = grp; j < wo; j += 4) {
            const int x = j * in_sx + dxs;
            if ((unsigned)x < (unsigned)wi) {
                sx += (ax * (float)x - 1.f) * scratch[(2 * r) * plane + (long long)j * co + c];
                sy += scratch[(2 * r + 1) * plane + (long long)j * co + c];
            }
        }
    }
    red[grp][0][cl] = sx; red[grp][1][cl] = sy;
    __syncthreads();
    if (grp != 0 || !ok) return;
    sx = (red[0][0][cl] + red[1][0][cl]) + (red[2][0][cl] + red[3][0][cl]);
    sy = (red[0][1][cl] + red[1][1][cl]) + (red[2][1][cl] + red[3][1][cl]);
    if (tap == kh * kw) { gb[c] = sx; return; }
    const int cin_v = ci_log + 2;
    gV[((long long)tap * cin_v + ci_log) * co + c] = sx;
    gV[((long long)tap * cin_v + ci_log + 1) * co + c] = sy;
}

// partial[blockIdx.y][c] = sum over this block's row range
template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ d, long long rows, int co, int ldo, int tx_n,
                                                      float* __restrict__ partial) {
    __shared__ float red[256];
    const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n, ty_n = 256 / tx_n;
    const int c = blockIdx.x * tx_n + tx;
    const long long per = (rows + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    float s = 0.f;
    if (c < co)
        for (long long r = r0 + ty; r < r1; r += ty_n) s += ld_as_float<T>(d + r * ldo + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && c < co) {
        for (int k = 1; k < ty_n; ++k) s += red[k * tx_n + tx];
        partial[(long long)blockIdx.y * co + c] = s;
    }
}

__global__ void col_sum_final_kernel(const float* __restrict__ partial, int nb, int co, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= co) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += partial[(long long)b * co + c];
    out[c] = s;
}

}  // namespace

extern "C" int ups_weight_prep(const float* src, int32_t ntaps, int32_t cin_v, int32_t ci_log, int32_t co, int32_t dtype,
                               void* w_fwd, int32_t ci_pad, void* w_dgrad, int32_t dgrad_rows, int32_t dgrad_k,
                               void* stream) {
    UPS_CHECK_ARG(src && (w_fwd || w_dgrad));
    UPS_CHECK_ARG(ci_log <= cin_v && (!w_fwd || ci_pad >= ci_log) && (!w_dgrad || dgrad_k >= co));
    const int bk = dtype == UPS_F32 ? 16 : 32;
    const long long total = (w_fwd ? (long long)ntaps * ups_cdiv(ci_pad, bk) * bk * co : 0) +
                            (w_dgrad ? (long long)ntaps * ups_cdiv(dgrad_k, bk) * bk * dgrad_rows : 0);
    int grid = ups_cdiv(total, 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32)
        hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(grid), dim3(256), 0, s, src, ntaps, cin_v, ci_log, co,
                           (float*)w_fwd, ci_pad, (float*)w_dgrad, dgrad_rows, dgrad_k);
    else if (dtype == UPS_F16)
        hipLaunchKernelGGL(weight_prep_kernel<f16>, dim3(grid), dim3(256), 0, s, src, ntaps, cin_v, ci_log, co,
                           (f16*)w_fwd, ci_pad, (f16*)w_dgrad, dgrad_rows, dgrad_k);
    else
        hipLaunchKernelGGL(weight_prep_kernel<bf16>, dim3(grid), dim3(256), 0, s, src, ntaps, cin_v, ci_log, co,
                           (bf16*)w_fwd, ci_pad, (bf16*)w_dgrad, dgrad_rows, dgrad_k);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

namespace {
// one block per row (output channel; transposed: input channel): row max, then the 64-byte pieces of that row for every
// (tap, 64-channel chunk).  Element (t, row, k) = V[t][k][row] (forward) or V[t][row][k] (transposed)
__global__ __launch_bounds__(256) void weight_prep_f8_kernel(const float* __restrict__ src, int ntaps, int cin_v, int rows, int kdim,
                                                             int co, int transpose, unsigned* __restrict__ wq,
                                                             float* __restrict__ deq) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int kc = (kdim + 63) / 64;
    auto at = [&](int t, int k) -> float {
        return transpose ? src[((long long)t * cin_v + c) * co + k] : src[((long long)t * cin_v + k) * co + c];
    };
    float m = 0.f;
    for (int i = tid; i < ntaps * kdim; i += 256) {
        const int t = i / kdim, k = i - t * kdim;
        m = fmaxf(m, fabsf(at(t, k)));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = m > 0.f ? 448.f / m : 1.f;
    if (tid == 0) deq[c] = m > 0.f ? m / 448.f : 1.f;
    for (int i = tid; i < ntaps * kc * 16; i += 256) {          // one dword (4 channels) per thread and step
        const int t = i / (kc * 16), r = i - t * (kc * 16), k = r >> 4, j4 = r & 15;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = k * 64 + j4 * 4 + e;
            f[e] = kk < kdim ? __builtin_amdgcn_fmed3f(at(t, kk) * sc, -448.f, 448.f) : 0.f;
        }
        int d = 0;
        d = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], d, false);
        d = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], d, true);
        wq[(((long long)t * kc + k) * rows + c) * 16 + j4] = (unsigned)d;
    }
}
}  // namespace

extern "C" int ups_weight_prep_f8(const float* src, int32_t ntaps, int32_t cin_v, int32_t ci_log, int32_t co, int32_t transpose,
                                  void* w_f8, float* deq, void* stream) {
    UPS_CHECK_ARG(src && w_f8 && deq && ntaps >= 1 && ci_log >= 1 && cin_v >= ci_log && co >= 1);
    const int rows = transpose ? ci_log : co, kdim = transpose ? co : ci_log;
    hipLaunchKernelGGL(weight_prep_f8_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, src, ntaps, cin_v, rows, kdim, co,
                       transpose, (unsigned*)w_f8, deq);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

namespace {
// one thread per (t9, k-chunk, row, 8-element piece) of the blocked-K weight image
__global__ __launch_bounds__(256) void weight_prep_d2s_kernel(const float* __restrict__ src, int cin_v, int ci_log, int co,
                                                              int pad_y, int pad_x, int C, int kc, uint4* __restrict__ w,
                                                              long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int piece = (int)(i & 3);
    long long r = i >> 2;
    const int row = (int)(r % (4 * C)); r /= 4 * C;
    const int k = (int)(r % kc);
    const int t9 = (int)(r / kc);
    const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
    const int cls = row / C, c = row - cls * C, py = cls >> 1, px = cls & 1;
    const int rr = py + pad_y - 2 * dy, ss = px + pad_x - 2 * dx;      // forward tap that maps lattice offset (dy, dx) to this class
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = k * 32 + piece * 8 + e;                        // gradient channel = forward output channel
        f[e] = (rr >= 0 && rr < 3 && ss >= 0 && ss < 3 && c < ci_log && kk < co)
                   ? src[((long long)(rr * 3 + ss) * cin_v + c) * co + kk] : 0.f;
    }
    w[i] = Chunk<bf16>::pack(f);
}
}  // namespace

extern "C" int ups_weight_prep_d2s(const float* src, int32_t cin_v, int32_t ci_log, int32_t co, int32_t pad_y, int32_t pad_x,
                                   int32_t C, void* w, void* stream) {
    UPS_CHECK_ARG(src && w && ci_log >= 1 && cin_v >= ci_log && co >= 1 && C >= ci_log && C >= 8 && (C & (C - 1)) == 0);
    const int kc = (co + 31) / 32;
    const long long total = 9ll * kc * 4 * C * 4;
    hipLaunchKernelGGL(weight_prep_d2s_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, cin_v,
                       ci_log, co, pad_y, pad_x, C, kc, (uint4*)w, total);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int64_t ups_prep_item_blocks(const ups_prep_item* item_host, int32_t dtype) {
    if (!item_host) return -1;
    return (prep_elems(*item_host, dtype == UPS_F32 ? 16 : 32) + 255) / 256;
}

extern "C" int ups_weight_prep_batch(const ups_prep_item* items, const int64_t* block_prefix, int32_t n_items,
                                     int64_t total_blocks, int32_t dtype, void* stream) {
    UPS_CHECK_ARG(items && block_prefix && n_items >= 1 && total_blocks >= 1 && total_blocks < 0x7fffffffLL);
    if (dtype == UPS_F32)
        hipLaunchKernelGGL(weight_prep_batch_kernel<float>, dim3((unsigned)ups_cdiv(total_blocks, PREP_CPB)), dim3(256), 0, (hipStream_t)stream,
                           items, (const long long*)block_prefix, n_items, (long long)total_blocks);
    else if (dtype == UPS_F16)
        hipLaunchKernelGGL(weight_prep_batch_kernel<f16>, dim3((unsigned)ups_cdiv(total_blocks, PREP_CPB)), dim3(256), 0, (hipStream_t)stream,
                           items, (const long long*)block_prefix, n_items, (long long)total_blocks);
    else
        hipLaunchKernelGGL(weight_prep_batch_kernel<bf16>, dim3((unsigned)ups_cdiv(total_blocks, PREP_CPB)), dim3(256), 0, (hipStream_t)stream,
                           items, (const long long*)block_prefix, n_items, (long long)total_blocks);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

static Taps3 make_taps3(int kh, int kw, const int32_t* dy, const int32_t* dx) {
    Taps3 t;
    for (int i = 0; i < 3; ++i) { t.dy[i] = 0; t.dx[i] = 0; }
    for (int r = 0; r < kh; ++r) t.dy[r] = dy[r * kw];
    for (int s = 0; s < kw; ++s) t.dx[s] = dx[s];
    return t;
}

extern "C" int ups_coord_table(const float* V, int32_t kh, int32_t kw, int32_t ci_log, int32_t co, const int32_t* tap_dy,
                               const int32_t* tap_dx, int32_t in_sy, int32_t in_sx, float ax, float ay, float* tab,
                               void* stream) {
    UPS_CHECK_ARG(V && tab && tap_dy && tap_dx && kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3);
    hipLaunchKernelGGL(coord_table_kernel, dim3(ups_cdiv(co, 128), 64), dim3(128), 0, (hipStream_t)stream, V, kh, kw,
                       ci_log, co, make_taps3(kh, kw, tap_dy, tap_dx), in_sy, in_sx, ax, ay, tab);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_batch_sum(const void* dout, int32_t dtype, int32_t n, int64_t pix, int32_t co, int32_t ldo, float* gsum,
                             void* stream) {
    UPS_CHECK_ARG(dout && gsum && n > 0 && pix > 0);
    UPS_CHECK_ARG(ldo % 8 == 0 && ((uintptr_t)dout & 15) == 0);
    int grid = ups_cdiv(pix * ups_cdiv(co, dtype == UPS_F32 ? 4 : 8), 256);
    if (grid > 16384) grid = 16384;
    if (dtype == UPS_F32)
        hipLaunchKernelGGL(batch_sum_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dout, n,
                           (long long)pix, co, ldo, gsum);
    else
        hipLaunchKernelGGL(batch_sum_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout, n,
                           (long long)pix, co, ldo, gsum);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_coord_wgrad(const float* gsum, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t co, int32_t kh,
                               int32_t kw, const int32_t* tap_dy, const int32_t* tap_dx, int32_t in_sy, int32_t in_sx,
                               float ax, float ay, int32_t ci_log, float* gradV, float* grad_bias, float* scratch,
                               void* stream) {
    UPS_CHECK_ARG(gsum && gradV && scratch && kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3);
    const Taps3 tp = make_taps3(kh, kw, tap_dy, tap_dx);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(coord_rows_kernel, dim3(ups_cdiv((long long)wo * co, 32)), dim3(256), 0, s, gsum, hi, ho, wo, co, kh,
                       tp, in_sy, ay, scratch);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(coord_cols_kernel, dim3(ups_cdiv(co, 64), kh * kw + 1), dim3(256), 0, s, scratch, wi, wo, co, kh, kw,
                       tp, in_sx, ax, ci_log, gradV, grad_bias);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}

extern "C" int ups_col_sum(const void* dout, int32_t dtype, int64_t rows, int32_t co, int32_t ldo, float* out,
                           float* workspace, void* stream) {
    UPS_CHECK_ARG(dout && out && workspace && rows > 0 && co > 0);
    int tx = 32;
    while (tx < co && tx < 256) tx *= 2;
    const int ty = 256 / tx;
    int nb = (int)((rows + (long long)ty * 16 - 1) / ((long long)ty * 16));
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UPS_F32)
        hipLaunchKernelGGL(col_sum_kernel<float>, dim3(ups_cdiv(co, tx), nb), dim3(256), 0, s, (const float*)dout,
                           (long long)rows, co, ldo, tx, workspace);
    else
        hipLaunchKernelGGL(col_sum_kernel<bf16>, dim3(ups_cdiv(co, tx), nb), dim3(256), 0, s, (const bf16*)dout,
                           (long long)rows, co, ldo, tx, workspace);
    UPS_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_sum_final_kernel, dim3(ups_cdiv(co, 256)), dim3(256), 0, s, workspace, nb, co, out);
    UPS_LAUNCH_CHECK();
    return UPS_OK;
}
