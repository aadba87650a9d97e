// First layers: 3x3 / stride-1 convolutions with at most 8 input channels (the RGB views: encoder_1 on the P x B part images
// -- `ea_in`, 640 images of 128x128 at the headline shape --, encoder_0, VGG block1_conv1; nn.py:617-664 with model.py:176-187
// for the part-masked form).  gfx950 only.
//
// These launches are output-write streams (ea_in: 16 MB of view + mask in, 671 MB out) that the general patch kernel runs as a
// chain of HBM round trips per block (halo patch -> barrier -> 3 tap-rows -> staged epilogue; 1.7-1.9 TB/s).  Here:
//   * im2col in the fragment addressing: K = 9 taps x 8 channels = 72 -> THREE v_mfma_f32_16x16x32_bf16 steps instead of nine; lane
//     (pixel column l & 15, k-group l >> 4) of step s reads the 16 bytes of tap 4 s + (l >> 4) of its pixel straight from global
//     memory (L1 / L2 hits: the input is tiny), no LDS image, no barrier;
//   * the whole weight set lives in registers (3 steps x NC output-channel groups x 4 VGPRs);
//   * part-masked form: a block loads a tile row's fragments and hard-mask words ONCE and produces the row of all P part images
//     from them (fragment & (bit p ? ~0 : 0)), i.e. the view is read once, not P times;
//   * every (tile row, part) result is one contiguous run of 16 pixels x co channels: transposed through a wave-private LDS row
//     (no block barrier anywhere in the kernel) and written with 16-byte stores.
// A block = 4 waves = one 16x16 tile of one view image; wave w owns tile rows 4 w .. 4 w + 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/upsparts_hip.h"
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

struct FirstK {
    const unsigned char* in; const unsigned char* w; unsigned char* out;
    unsigned char* sign_out;            // ups_conv_desc.sign_out: one sign byte beside every stored 16-byte chunk (or NULL)
    const float* bias; const float* coord_tab; const unsigned* mask;
    int B, P, h, wd, ldi, co, co_fill, ldo, out_act, tiles_x, tiles_y;
    float slope;
};

template <int NC>       // groups of 16 output channels: 2 (co_fill 32) or 4 (co_fill 64)
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const FirstK p) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][16 * NC * 32];     // per wave: 16 pixels x NC*16 channels x 2 B
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p16 = lane & 15, q16 = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int tx0 = tx * 16, ty0 = ty * 16;
    const float oact_ns = ups_slope_eff(p.out_act, p.slope);

    // weights: B operand of step s, channel group j = rows 16 j + p16, tap 4 s + q16, channels 0..7 of the blocked-K layout
    // [tap][1 chunk][co][32] (ups_weight_prep); taps 9..11 of the last step are zero
    bf16x8 wf[3][NC];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int tap = 4 * s + q16, c = 16 * j + p16;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (tap < 9 && c < p.co) v = *(const uint4*)(p.w + ((long long)tap * p.co + c) * 64);
            __builtin_memcpy(&wf[s][j], &v, 16);
        }
    // per lane: the output channels 16 j + 4 q16 + e; bias + CoordConv terms of an interior pixel (class 63): affine in (x, y)
    const int x = tx0 + p16;
    const bool xin = x > 0 && x + 1 < p.wd;
    float addx[NC][4], ty2[NC][4];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 16 * j + 4 * q16 + e;
            const bool cv = col < p.co;
            float v = (cv && p.bias) ? p.bias[col] : 0.f;
            ty2[j][e] = 0.f;
            if (cv && p.coord_tab) {
                const float* tb = p.coord_tab + (long long)63 * 3 * p.co + col;
                v += tb[0] + (float)x * tb[p.co];
                ty2[j][e] = tb[2 * p.co];
            }
            addx[j][e] = v;
        }
    const unsigned char* inb = p.in + (long long)b * p.h * p.wd * p.ldi * 2;
    const unsigned* mb = p.mask ? p.mask + (long long)b * p.h * p.wd : nullptr;
    unsigned char* st = stage[wid];
    // the A fragments of the wave's four rows (and the hard-mask words of their source pixels): all twelve loads in flight at once
    uint4 afr[4][3];
    unsigned mwr[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + q16;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int ys = ty0 + 4 * wid + i + dy, xs = x + dx;
            const bool ok = tap < 9 && (unsigned)ys < (unsigned)p.h && (unsigned)xs < (unsigned)p.wd;
            const long long pix = ok ? (long long)ys * p.wd + xs : 0;
            uint4 v = *(const uint4*)(inb + pix * p.ldi * 2);
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            afr[i][s] = v;
            mwr[i][s] = mb ? mb[pix] : 0xffffffffu;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = ty0 + 4 * wid + i;
        const uint4 (&af)[3] = afr[i];
        const unsigned (&mw)[3] = mwr[i];
        // bias + CoordConv term of this row's pixel (border pixels: class table, as conv3x3_patch.hip's epilogue)
        float addv[NC][4];
        const bool yin = y > 0 && y + 1 < p.h;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) addv[j][e] = addx[j][e] + (float)y * ty2[j][e];
        if (p.coord_tab && !(xin && yin)) {
            const int ym = (y > 0 ? 1 : 0) | 2 | (y + 1 < p.h ? 4 : 0), xm = (x > 0 ? 1 : 0) | 2 | (x + 1 < p.wd ? 4 : 0);
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = 16 * j + 4 * q16 + e;
                    if (col < p.co) {
                        const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col;
                        addv[j][e] = (p.bias ? p.bias[col] : 0.f) + (tb[0] + (float)x * tb[p.co] + (float)y * tb[2 * p.co]);
                    }
                }
        }
#pragma unroll 1
        for (int part = 0; part < p.P; ++part) {
            f32x4v acc[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const unsigned keep = ((mw[s] >> part) & 1u) ? 0xffffffffu : 0u;
                const uint4 m = make_uint4(af[s].x & keep, af[s].y & keep, af[s].z & keep, af[s].w & keep);
                bf16x8 a;
                __builtin_memcpy(&a, &m, 16);
#pragma unroll
                for (int j = 0; j < NC; ++j)       // weights as the row operand: a lane holds 4 consecutive channels of its pixel
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], a, acc[j], 0, 0, 0);
            }
            // epilogue: + bias / CoordConv, stored activation, bf16; lane (p16, q16) -> 8 bytes at [pixel p16][channel 16 j + 4 q16]
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[j][e] + addv[j][e];
                    if (p.out_act) v[e] = ups_vmax(v[e], oact_ns * v[e]);
                    if (16 * j + 4 * q16 + e >= p.co) v[e] = 0.f;
                }
                *(uint2*)(st + p16 * (NC * 32) + j * 32 + q16 * 8) = make_uint2(Chunk<bf16>::pk(v[0], v[1]), Chunk<bf16>::pk(v[2], v[3]));
            }
            // the row back out: 16 pixels x co_fill channels, 16 bytes per lane and store, coalesced
            unsigned char* orow = p.out + (((long long)(part * p.B + b) * p.h + y) * p.wd + tx0) * p.ldo * 2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NC / 2; ++k) {
                const int idx = lane + 64 * k;                 // 16-byte piece of the staged row
                const int px = idx / (NC * 2), ch = idx - px * (NC * 2);
                if (ch * 8 < p.co_fill) {
                    const uint4 o = *(const uint4*)(st + idx * 16);
                    *(uint4*)(orow + (long long)px * p.ldo * 2 + ch * 16) = o;
                    if (p.sign_out)
                        p.sign_out[((((long long)(part * p.B + b) * p.h + y) * p.wd + tx0) + px) * (p.ldo >> 3) + ch] = (unsigned char)ups_sign_byte(o);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging row is rewritten by the next part)
        }
    }
}

}  // namespace

// Internal entry of ups_conv_igemm's dispatcher.  Returns 1 if the problem is not a first-layer problem, 0 when launched.
int ups_conv3x3_first_try(const ups_conv_desc* d, hipStream_t s) {
    { const char* e = getenv("UPS_FIRST_LAYER"); if (e && e[0] == '0') return 1; }     // (read per call: the parity test toggles it)
    if (d->dtype != UPS_BF16 || d->ntaps != 9 || d->ci != 8 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 ||
        d->out_oy || d->out_ox || d->hi != d->ho || d->wi != d->wo || d->out_h != d->ho || d->out_w != d->wo)
        return 1;
    if ((d->hi % 16) || (d->wi % 16) || d->act_in != UPS_ACT_NONE || d->res || d->dact || d->d2s || d->out_f32 || d->mask_grad ||
        d->f8_deq || d->in_f8 || d->out_f8 || d->out_f8_amax || d->res_act)
        return 1;
    if (!(d->co_fill == 32 || d->co_fill == 64) || (d->ldo & 7) || d->co > d->co_fill) return 1;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1 || d->tap_w[t] != t) return 1;
    FirstK k;
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.out = (unsigned char*)d->out;
    k.bias = d->bias; k.coord_tab = d->coord_tab; k.mask = d->mask_bits;
    k.sign_out = (unsigned char*)d->sign_out;
    k.B = d->n; k.P = 1;
    if (d->mask_bits) {
        if (d->mask_batch <= 0 || d->n % d->mask_batch) return 1;
        k.B = d->mask_batch; k.P = d->n / d->mask_batch;
        if (k.P > 32) return 1;
    }
    k.h = d->hi; k.wd = d->wi; k.ldi = d->ldi; k.co = d->co; k.co_fill = d->co_fill; k.ldo = d->ldo;
    k.out_act = d->out_act; k.slope = d->act_slope;
    k.tiles_x = d->wi / 16; k.tiles_y = d->hi / 16;
    const long long blocks = (long long)k.B * k.tiles_x * k.tiles_y;
    if (blocks >= (1ll << 31)) return 1;
    if (d->co_fill == 32) hipLaunchKernelGGL((conv3x3_first_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((conv3x3_first_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, k);
    if (k.sign_out) g_ups_sign_written = 1;
    return 0;
}

