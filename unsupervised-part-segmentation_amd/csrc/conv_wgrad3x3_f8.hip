// fp8 weight gradient of the wide 3x3 / stride-1 'SAME' convolutions (BASELINE config #5, nn.py:661-663; round 5):
//     dV[tap][ci][co] = sum_{img,y,x} act(in)[img][y+dy][x+dx][ci] * dout[img][y][x][co]
// on the block-scaled K = 128 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, E8M0 block scales 1.0 -- the per-tensor scales are
// divided out when the slab is written): e4m3 activations x e5m2 gradients, fp32 accumulation.
//
// Pixels are the GEMM K.  A unit is a half-tile of 8 rows x 16 pixels = 128 pixels = ONE K-step: lane l of a fragment holds
// row / column l & 15 (an input channel of X^T, an output channel of dout) and the 32 pixels of tile rows 2q, 2q + 1 (q = l >> 4),
// fetched from the pixel-major LDS images with four transposing reads (ds_read_b64_tr_b8: the 16 lanes of a group address
// 8 pixels x 16 one-byte channels -- lane i: pixel i >> 1, 8-byte piece i & 1 -- and lane i receives channel i of the 8 pixels;
// layout probed on the hardware, tools/probes/tr8_probe.hip).  The order of k is free as long as both operands agree, and they
// do: the same (q, read, pixel) walk on both images, the X window shifted by the tap.
//
// Block = 512 threads = 8 waves = 4 (16 input channels) x 2 (64 output channels); it owns a 64 x 128 slice of dV for ALL nine
// taps (9 x 1 x 4 accumulator tiles of 16 x 16 = 144 registers per lane, as the bf16 kernel) and walks its share of the units:
//   dout   the e5m2 copy its producer wrote (ups_wgrad_desc.dout_f8: the input-gradient epilogue of the layer above / the
//          bilinear backward kernel; the same copy the layer's input-gradient launch reads) arrives by LDS-DMA into an unpadded
//          [8 rows][16 px][128 B] tile whose 16-byte slots are XOR-swizzled on the source side (slot ^= ((x >> 1) & 3) | ((r >> 1) & 1) << 2:
//          the two 16-lane groups of an LDS cycle read 8 + 8 pixels at rows r, r + 2 -- sixteen different bank quads);
//   X      the 10 x 18 halo patch of the bf16 / fp16 forward input is quantised while it is staged (activation-on-load, running
//          maximum for the next step's delayed scale, * scale, v_cvt_pk_fp8_f32) into [10 rows][20 px][80 B]: pixel pitch 80 B =
//          5 bank quads, row pitch 100 quads -- 8 consecutive pixels hit 8 different quads (5 x mod 16) and the row pair two
//          rows down hits the other 8 (2 x 100 = 8 mod 16) for EVERY tap shift, so the nine windows are immediate offsets from
//          one lane address and there is no swizzle to undo.
// Two X images, three dout images (staging pipelined two units ahead); one barrier per unit (36 MFMAs of 32 cycles per wave).  The bias gradient is one more MFMA per
// output block against an all-ones e4m3 operand (waves of input-channel half 0 of the first input-channel tile).
// Slabs, split-K and the deterministic reduce are the bf16 kernel's (conv_wgrad3x3.hip / conv_wgrad.hip).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TW = 16, TH = 8, PWID = TW + 2, PROWS = TH + 2;
constexpr int CB = 64, BN = 128;
constexpr int XPP = 80, XRP = 20 * XPP, XB = PROWS * XRP;      // X image: 16 000 B
constexpr int DPP = BN, DB = TH * TW * DPP;                     // dout image: 16 384 B
constexpr int NITEMS = PROWS * PWID * (CB / 16);                // 720 staged items of 16 channels
constexpr int NX = (NITEMS + 511) / 512;                        // 2 per thread

struct Wg8K {
    int n, h, w, ci, ldi, ci_log, cin_v, co, ldo, ldo8, want_bias, units_total, units_per, tiles_x, tiles_y, in_f16, act_in;
    float act_slope;
    unsigned long long tap_wi;
    const void* in; const unsigned char* dout8; float* ws;
    const float* sx; const float* sg; float* amax;
};

typedef __attribute__((ext_vector_type(2))) int i32x2v;
typedef __attribute__((ext_vector_type(8))) int i32x8v;
typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((address_space(3))) i32x2v lds_i32x2v;

__device__ __forceinline__ int g_w8(unsigned long long wi, int t) { return (int)((wi >> (4 * t)) & 15); }

// four transposing reads = the 32 k-values (pixels of tile rows 2q, 2q + 1) of one operand fragment; `a` = lane address of
// (row 2q, pixel p = (l & 15) >> 1, piece l & 1), rp / hp = byte pitch of a tile row / of 8 pixels
template <int RP, int HP>
__device__ __forceinline__ i32x8v frag8(const unsigned char* a) {
    const i32x2v r0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2v*)(a));
    const i32x2v r1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2v*)(a + HP));
    const i32x2v r2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2v*)(a + RP));
    const i32x2v r3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2v*)(a + RP + HP));
    return (i32x8v){r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
}

// 16 bf16 / fp16 values (two 16-byte words) -> act -> running max -> * scale -> 16 e4m3 bytes
template <bool F16IN, bool ACT>
__device__ __forceinline__ uint4 quant16(uint4 u0, uint4 u1, float sc, float ns, float& amax) {
    const unsigned w[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
    int d[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float lo, hi;
        if constexpr (F16IN) ups_unpack2<f16>(w[k], lo, hi);
        else { lo = __uint_as_float(w[k] << 16); hi = __uint_as_float(w[k] & 0xffff0000u); }
        if constexpr (ACT) { lo = ups_vmax(lo, ns * lo); hi = ups_vmax(hi, ns * hi); }
        amax = fmaxf(fmaxf(amax, fabsf(lo)), fabsf(hi));
        lo = __builtin_amdgcn_fmed3f(lo * sc, -448.f, 448.f);
        hi = __builtin_amdgcn_fmed3f(hi * sc, -448.f, 448.f);
        if (k & 1) d[k >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, d[k >> 1], true);
        else d[k >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, d[k >> 1], false);
    }
    return make_uint4((unsigned)d[0], (unsigned)d[1], (unsigned)d[2], (unsigned)d[3]);
}

// F16IN: `in` holds fp16 (the mask decoder's forward tensors); ACT: activation-on-load (instances: straight-line quantisation code)
template <bool F16IN, bool ACT>
__global__ __launch_bounds__(512) void conv_wgrad3x3_f8_kernel(const Wg8K p, const int cit, const int cot, const int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xbuf = smem;                 // 2 x XB
    unsigned char* Dbuf = smem + 2 * XB;        // 3 x DB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float act_ns = ups_slope_eff(p.act_in, p.act_slope);
    const float sx = *p.sx, sg = *p.sg;
    // XCD-aware order (as the bf16 kernel): the cit * cot blocks of one K split share an XCD's L2
    const int pairs = cit * cot;
    int pair = blockIdx.x % pairs, split = blockIdx.x / pairs;
    if ((nsplit & 7) == 0) {
        const int j = blockIdx.x >> 3;
        pair = j % pairs;
        split = (j / pairs) * 8 + (blockIdx.x & 7);
    }
    const int cot_i = pair % cot, cit_i = pair / cot;
    const int w_ci = wid & 3, w_co = wid >> 2;              // wave tile: 16 input channels x 64 output channels
    const int ci0 = cit_i * CB, co0 = cot_i * BN;
    // bias gradient: the waves of input-channel quarters 0 / 1 of the first input-channel tile take output blocks 0, 1 / 2, 3 of
    // their 64 channels (two accumulator tiles per wave instead of four: registers)
    const bool do_bias = p.want_bias && cit_i == 0 && w_ci < 2;
    const int u_begin = split * p.units_per;
    const int u_end = min(p.units_total, u_begin + p.units_per);

    // ---- staging tables (per thread, unit-independent)
    unsigned xr[NX], xs[NX];        // image-pitch pixel index of the item relative to the patch origin; LDS offset | edge flags << 16
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int item = min(tid + 512 * k, NITEMS - 1);
        const int pix = item >> 2, sl = item & 3;
        const int py = pix / PWID, px = pix - py * PWID;
        unsigned fl = (py == 0 ? 1u : 0u) | (py == PROWS - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PWID - 1 ? 8u : 0u);
        if (tid + 512 * k >= NITEMS) fl |= 16u;
        xr[k] = (unsigned)(py * p.w + px);
        xs[k] = (unsigned)(py * XRP + px * XPP + sl * 16) | (fl << 16);
    }
    const unsigned x_rowb = (unsigned)p.ldi * 2u;
    const unsigned x_chb = (unsigned)(ci0 + (tid & 3) * 16) * 2u;        // 512 % 4 == 0: one channel slot per thread
    unsigned dd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = wid + 8 * k;                                         // 1 KiB piece: tile row j >> 1, pixels 8 (j & 1) ..
        const int r = j >> 1, x = (j & 1) * 8 + (lane >> 3), sl = lane & 7;
        const int c = sl ^ (((x >> 1) & 3) | (((r >> 1) & 1) << 2));
        dd[k] = (unsigned)((r * p.w + x) * p.ldo8 + co0 + c * 16);
    }
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    auto unit_origin = [&](int u, int& img, int& y0, int& x0) {
        const int half = u & 1;
        int t = u >> 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        img = t / p.tiles_y;
        y0 = ty * 16 + half * TH; x0 = tx * TW;
    };
    // X loads of unit u into registers (plain loads: hipcc keeps their waits and register hazards right).  The dout DMA of the SAME
    // unit is inline asm (the builtin would put an lgkmcnt(0) in front of every later LDS access) and therefore invisible to
    // hipcc's wait-count pass -- it is always issued right BEFORE these loads, so the vmcnt wait the compiler places in front of
    // their first use covers it too (vector-memory operations complete in issue order).
    uint4 rx[NX][2];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    auto dma_d = [&](int u, int dbuf) __attribute__((always_inline)) {
        int img, y0, x0;
        unit_origin(u, img, y0, x0);
        const unsigned char* db = p.dout8 + (((long long)img * p.h + y0) * p.w + x0) * p.ldo8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(2 * XB + dbuf * DB + (wid + 8 * k) * 1024));
#if !defined(UPS_W8_NO_DMA)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lds_dst), "v"(dd[k]), "s"(db) : "memory", "m0");
#else
            asm volatile("" :: "s"(lds_dst), "v"(dd[k]), "s"(db));                                  // (ablation: no dout traffic)
#endif
        }
    };
    auto load_x = [&](int u) __attribute__((always_inline)) {
        int img, y0, x0;
        unit_origin(u, img, y0, x0);
        const unsigned edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.h ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.w ? 8u : 0u) | 16u;
        const long long org = ((long long)img * p.h + (y0 - 1)) * p.w + (x0 - 1);
        const unsigned char* xb = (const unsigned char*)p.in + org * p.ldi * 2;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            unsigned r = xr[k], f = xs[k];
            asm volatile("" : "+v"(r), "+v"(f));
            uint4 v0 = zero4, v1 = zero4;
#if !defined(UPS_W8_NO_XLOAD)
            if (((f >> 16) & edge) == 0u) {
                const unsigned char* src = xb + (__umul24(r, x_rowb) + x_chb);
                v0 = *(const uint4*)src; v1 = *(const uint4*)(src + 16);
            }
#else
            v0 = v1 = make_uint4(r, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);                          // (ablation: no X traffic)
#endif
            rx[k][0] = v0; rx[k][1] = v1;
        }
    };
    float amax = 0.f;
    auto store_x = [&](int buf, int k) __attribute__((always_inline)) {
        unsigned char* X = Xbuf + buf * XB;
        unsigned f = xs[k];
        asm volatile("" : "+v"(f));
#if defined(UPS_W8_NO_QUANT)
        const uint4 v = rx[k][0];                                                                     // (ablation: no conversion arithmetic)
#else
        const uint4 v = quant16<F16IN, ACT>(rx[k][0], rx[k][1], sx, act_ns, amax);
#endif
        if (k + 1 < NX || tid + 512 * k < NITEMS) *(uint4*)(X + (f & 0xffffu)) = v;
    };
    static_assert(NX == 2, "pipelined staging: two items per thread and unit");

    f32x4v acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[t][b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    f32x4v accb[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
    const i32x8v ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};   // e4m3 1.0

    // ---- fragment lane addresses
    const int q = lane >> 4, pp = (lane & 15) >> 1, piece = lane & 1;
    const int xa = (2 * q) * XRP + pp * XPP + piece * 8 + w_ci * 16;                 // + tap window
    const int gd = ((pp >> 1) & 3) | ((q & 1) << 2);
    int da[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) da[jb] = ((2 * q) * TW + pp) * DPP + (((w_co * 4 + jb) ^ gd) << 4) + piece * 8;

    // Software-pipelined staging (one block per CU runs its eight waves in lock-step between barriers: a load -> wait -> convert ->
    // ds_write phase at the end of every unit would idle the matrix pipe for its whole length -- and on the block-scaled MFMA a
    // unit's 36 MFMAs are only ~1.1 k cycles).  While unit u is multiplied: [taps 3, 5] the X items of unit u+1 (requested during
    // unit u-1) are converted and written to the other X image, then the dout DMA and the X loads of unit u+2 are issued -- a
    // whole unit of latency cover for both.  dout images: THREE (the DMA of unit u+2 lands while unit u+1's image is waiting and
    // unit u's is being read); X images: two.
    if (u_begin < u_end) {
        dma_d(u_begin, 0); load_x(u_begin);
        store_x(0, 0); store_x(0, 1);
        if (u_begin + 1 < u_end) { dma_d(u_begin + 1, 1); load_x(u_begin + 1); }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int dcur = 0;                           // dout image of the running unit (u - u_begin) % 3
    for (int u = u_begin; u < u_end; ++u) {
        const int buf = (u - u_begin) & 1;
        const int dnext2 = dcur == 0 ? 2 : dcur - 1;      // (dcur + 2) % 3
        const unsigned char* X = Xbuf + buf * XB + xa;
        const unsigned char* D = Dbuf + dcur * DB;
        // a wave's tile is 16 x 64 (one X fragment per tap against four dout fragments that stay resident for the unit): 36 + 16
        // transposing reads per 36 MFMAs (the 32 x 32 form read 72 + 8: with the matrix work halved by the fp8 MFMA the LDS reads
        // and their issue slots were what the loop was short of -- the same launch without MFMAs took 1.0 of its 1.4 ms)
        i32x8v fb[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) fb[jb] = frag8<TW * DPP, 8 * DPP>(D + da[jb]);
        i32x8v fa[2];
        fa[0] = frag8<XRP, 8 * XPP>(X);                                                          // tap 0: window (0, 0)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) {
                const int off = ((t + 1) / 3) * XRP + ((t + 1) % 3) * XPP;                        // taps r-major: dy = t / 3 - 1, dx = t % 3 - 1
                fa[(t + 1) & 1] = frag8<XRP, 8 * XPP>(X + off);
            }
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#if defined(UPS_W8_NO_MFMA)
                acc[t][jb][0] += __int_as_float(fa[t & 1][jb] ^ fb[jb][t & 7]);                           // (ablation: no matrix work)
#else
                acc[t][jb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[t & 1], fb[jb], acc[t][jb], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) asm volatile("" : "+v"(acc[t][jb]));
            __builtin_amdgcn_sched_barrier(0);
            if (t == 3 && u + 1 < u_end) {
                store_x(buf ^ 1, 0);                // (the compiler's vmcnt wait in front of this also covers the DMA of unit u+1)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t == 5 && u + 1 < u_end) {
                store_x(buf ^ 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 2 < u_end) { dma_d(u + 2, dnext2); load_x(u + 2); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (do_bias) {
            if (w_ci == 0) {
                accb[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, fb[0], accb[0], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                accb[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, fb[1], accb[1], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else {
                accb[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, fb[2], accb[0], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                accb[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, fb[3], accb[1], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
        dcur = dcur == 2 ? 0 : dcur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- next step's activation scale
    {
        const float m = wave_max(amax);
        if (lane == 0 && p.amax) ups_amax_slot(p.amax + (blockIdx.x & 63), m);
    }
    // ---- the block's slab: rows < ci_log of every tap, dequantised
    const float inv = 1.f / (sx * sg);
    const long long slab_sz = (long long)9 * p.cin_v * p.co + p.co;
    float* slab = p.ws + (long long)split * slab_sz;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tw = g_w8(p.tap_wi, t);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int col = co0 + w_co * 64 + jb * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ci0 + w_ci * 16 + 4 * (lane >> 4) + e;
                if (row < p.ci_log && col < p.co) slab[((long long)tw * p.cin_v + row) * p.co + col] = acc[t][jb][e] * inv;
            }
        }
    }
    if (do_bias && (lane >> 4) == 0) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int col = co0 + w_co * 64 + (2 * w_ci + jb) * 16 + lane;
            if (col < p.co) slab[(long long)9 * p.cin_v * p.co + col] = accb[jb][0] * (1.f / sg);
        }
    }
}

bool eligible8(const ups_wgrad_desc* d) {
    if (!d->dout_f8 || d->dtype != UPS_BF16 || d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1) return false;
    if (d->hi != d->ho || d->wi != d->wo || d->hi % 16 || d->wi % 16 || d->mask_bits) return false;
    if (d->ci % CB || d->co % BN || d->ldo % 16) return false;
    if (19ll * d->wi * d->ldi * 2 >= (1ll << 31) || 17ll * d->wi * d->ldo >= (1ll << 31)) return false;
    if (18ll * d->wi >= (1 << 24) || (long long)d->ldi * 2 >= (1 << 24)) return false;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1 || d->tap_w[t] < 0 || d->tap_w[t] > 8) return false;
    return true;
}

}  // namespace

// Returns 1 when the problem is not one of these; otherwise the block split count (slabs = splits).
int ups_wgrad3x3_f8_plan(const ups_wgrad_desc* d, int* splitk, int* slabs) {
    if (!eligible8(d)) return 1;
    const int pairs = (d->ci / CB) * (d->co / BN);
    const int units = d->n * (d->hi / 16) * (d->wi / 16) * 2;
    int sk = ups_cdiv(256, pairs);                  // one block per CU (81 KB of LDS, 2 waves per SIMD on 144 accumulator registers)
    if (sk > units / 4) sk = units / 4 > 0 ? units / 4 : 1;
    if (sk > 256) sk = 256;
    if (sk >= 8) sk &= ~7;
    *splitk = sk;
    *slabs = sk;
    return 0;
}

int ups_wgrad3x3_f8_run(const ups_wgrad_desc* d, hipStream_t s) {
    if (!d->dout_f8_scale || !d->in_f8_scale) return UPS_E_ARG;
    Wg8K k;
    k.n = d->n; k.h = d->hi; k.w = d->wi; k.ci = d->ci; k.ldi = d->ldi; k.ci_log = d->ci_log; k.cin_v = d->cin_v;
    k.co = d->co; k.ldo = d->ldo; k.ldo8 = d->ldo; k.act_in = d->act_in; k.act_slope = d->act_slope; k.want_bias = d->grad_bias != nullptr;
    k.tiles_x = d->wi / 16; k.tiles_y = d->hi / 16;
    k.units_total = d->n * k.tiles_x * k.tiles_y * 2;
    k.units_per = ups_cdiv(k.units_total, d->splitk);
    k.in = d->in; k.dout8 = (const unsigned char*)d->dout_f8; k.ws = d->workspace;
    k.sx = d->in_f8_scale; k.sg = d->dout_f8_scale; k.amax = d->in_f8_amax;
    k.in_f16 = d->in_f16;
    k.tap_wi = 0;
    for (int t = 0; t < 9; ++t) k.tap_wi |= (unsigned long long)d->tap_w[t] << (4 * t);
    const int cit = d->ci / CB, cot = d->co / BN;
    constexpr size_t shmem = 2 * (size_t)XB + 3 * (size_t)DB;
#define UPS_W8_LAUNCH(F16V, ACTV)                                                                                                   \
    do {                                                                                                                              \
        static UpsPerDevice attr_set;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            if (hipFuncSetAttribute((const void*)conv_wgrad3x3_f8_kernel<F16V, ACTV>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                    (int)shmem) != hipSuccess) return UPS_E_LAUNCH;                                                   \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((conv_wgrad3x3_f8_kernel<F16V, ACTV>), dim3(cit * cot * d->splitk), dim3(512), shmem, s, k, cit, cot, d->splitk); \
    } while (0)
    const bool act = d->act_in != UPS_ACT_NONE;
    if (d->in_f16) { if (act) UPS_W8_LAUNCH(true, true); else UPS_W8_LAUNCH(true, false); }
    else { if (act) UPS_W8_LAUNCH(false, true); else UPS_W8_LAUNCH(false, false); }
#undef UPS_W8_LAUNCH
    return UPS_OK;
}
