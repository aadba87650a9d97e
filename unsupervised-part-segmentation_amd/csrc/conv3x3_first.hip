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

// RW (round 6, late): rows per wave.  A block is a strip of 16 columns x 4 RW rows of one view image; a wave walks its RW rows with the
// fragments of the next two rows in flight (a ring of three register sets) instead of requesting four rows at once and sitting out the
// round trip: with 148-220 registers two or three blocks share a CU, and a 16 x 16 tile per block left ~1 TB/s of stores behind one
// exposed round trip per tile.
template <int NC, int RW>       // NC: groups of 16 output channels: 2 (co_fill 32) or 4 (co_fill 64)
// (132-236 registers: two or three waves per SIMD.  Occupancy bounds of 4 / 5 waves spill 26-188 registers and run 1.5-4x slower.)
__global__ __launch_bounds__(256) void conv3x3_first_kernel(const FirstK p) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][16 * NC * 32];     // per wave: 16 pixels x NC*16 channels x 2 B
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform: the row bases below stay on the scalar unit)
    const int p16 = lane & 15, q16 = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y; const int b = t / p.tiles_y;
    const int tx0 = tx * 16, ty0 = ty * (4 * RW);
    const float oact_ns = p.out_act ? ups_slope_eff(p.out_act, p.slope) : 1.f;      // stored value = max(v, ns v): ns = 1 leaves v as it is (no branch)

    // weights: B operand of step s, channel group j = rows 16 j + p16, tap 4 s + q16, channels 0..7 of the blocked-K layout
    // [tap][1 chunk][co][32] (ups_weight_prep); taps 9..11 of the last step are zero
    bf16x8 wf[3][NC];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            // (every load of this kernel is UNCONDITIONAL, from a clamped address, and masked afterwards: a load under a branch makes
            // hipcc wait vmcnt(0) behind it, and the listing had every fragment load followed by its own full round trip)
            const int tap = 4 * s + q16, c = 16 * j + p16;
            uint4 v = *(const uint4*)(p.w + ((long long)min(tap, 8) * p.co + min(c, p.co - 1)) * 64);
            const unsigned wk = (tap < 9 && c < p.co) ? 0xffffffffu : 0u;
            v = make_uint4(v.x & wk, v.y & wk, v.z & wk, v.w & wk);
            __builtin_memcpy(&wf[s][j], &v, 16);
        }
    // per lane: the output channels 16 j + 4 q16 + e; bias + CoordConv terms of a pixel of the lane's COLUMN class in an interior row
    // (class 7 * 8 + xm; 63 for an interior column): affine in (x, y).  Round 6, late: until then the terms were the interior class's and
    // every pixel of a first / last column took the per-row table path below -- 32 dependent dword loads per row, in all rows of the 28
    // border tiles of 64: those blocks ran ~6x longer than interior ones and the launches sat at 0.9-1.4 TB/s written.  Now only the first
    // and the last ROW of the image take that path.
    const int x = tx0 + p16;
    const int xm_l = (x > 0 ? 1 : 0) | 2 | (x + 1 < p.wd ? 4 : 0);
    float addx[NC][4], ty2[NC][4];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 16 * j + 4 * q16 + e, colc = min(col, p.co - 1);
            const bool cv = col < p.co;
            const float* bp = p.bias ? p.bias : (const float*)p.w;                         // (a valid address either way; masked below)
            const float* tb = (p.coord_tab ? p.coord_tab + (long long)(7 * 8 + xm_l) * 3 * p.co : (const float*)p.w) + colc;
            const float bv = bp[colc], t0 = tb[0], t1 = tb[p.coord_tab ? p.co : 0], t2 = tb[p.coord_tab ? 2 * p.co : 0];
            float v = (cv && p.bias) ? bv : 0.f;
            if (cv && p.coord_tab) v += t0 + (float)x * t1;
            ty2[j][e] = (cv && p.coord_tab) ? t2 : 0.f;
            addx[j][e] = v;
        }
    const unsigned char* inb = p.in + (long long)b * p.h * p.wd * p.ldi * 2;
    const unsigned* mb = p.mask ? p.mask + (long long)b * p.h * p.wd : nullptr;
    const unsigned* mbp = mb ? mb : (const unsigned*)(p.in + (long long)b * p.h * p.wd * p.ldi * 2);      // (loaded unconditionally, dropped without a mask)
    unsigned char* st = stage[wid];
    unsigned st_off[NC / 2], sg_off[NC / 2];          // the lane's 16-byte pieces of a staged row: byte offset in the output row / its sign byte
    bool st_on[NC / 2];
#pragma unroll
    for (int k = 0; k < NC / 2; ++k) {
        const int idx = lane + 64 * k, px = idx / (NC * 2), ch = idx - px * (NC * 2);
        st_on[k] = ch * 8 < p.co_fill;
        st_off[k] = (unsigned)(px * p.ldo * 2 + ch * 16);
        sg_off[k] = (unsigned)(px * (p.ldo >> 3) + ch);
    }
    // the A fragments of a row (and the hard-mask words of their source pixels): three loads per lane; rows i + 1, i + 2 in flight under row i
    // (ring of RS register sets, RS - 1 rows ahead: vmcnt retires in issue order, so the wait for row i + 1's fragments also waits for every
    // STORE issued before them -- with two rows ahead that was the previous row's store, a full write round trip per row; four rows ahead the
    // stores in front of the awaited loads are four rows old)
    constexpr int RS = RW >= 8 ? 5 : 3;
    uint4 afr[RS][3];
    unsigned mwr[RS][3];
    auto issue = [&](int i, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + q16;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int ys = ty0 + RW * wid + i + dy, xs = x + dx;
            const bool ok = tap < 9 && (unsigned)ys < (unsigned)p.h && (unsigned)xs < (unsigned)p.wd;
            const unsigned pix = ok ? (unsigned)(ys * p.wd + xs) : 0u;          // (an image is < 2^31 bytes: uniform base + 32-bit lane offset)
            uint4 v = *(const uint4*)(inb + pix * (unsigned)(p.ldi * 2));
            const unsigned ak = ok ? 0xffffffffu : 0u;          // (an AND, not a select under a branch: hipcc waits vmcnt(0) inside such a branch)
            afr[slot][s] = make_uint4(v.x & ak, v.y & ak, v.z & ak, v.w & ak);
            const unsigned mwv = mbp[pix];
            mwr[slot][s] = mb ? mwv : 0xffffffffu;
        }
    };
#pragma unroll
    for (int i = 0; i < RS - 1; ++i)
        if (i < RW) issue(i, i);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        if (i + RS - 1 < RW) issue(i + RS - 1, (i + RS - 1) % RS);
        const int y = ty0 + RW * wid + i;
        const uint4 (&af)[3] = afr[i % RS];
        const unsigned (&mw)[3] = mwr[i % RS];
        // bias + CoordConv term of this row's pixel (border pixels: class table, as conv3x3_patch.hip's epilogue)
        float addv[NC][4];
        const bool yin = y > 0 && y + 1 < p.h;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) addv[j][e] = addx[j][e] + (float)y * ty2[j][e];
        if (p.coord_tab && !yin) {        // (wave-uniform: y is the wave's row)
            const int ym = (y > 0 ? 1 : 0) | 2 | (y + 1 < p.h ? 4 : 0), xm = xm_l;
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
                    v[e] = ups_vmax(v[e], oact_ns * v[e]);
                    if (16 * j + 4 * q16 + e >= p.co) v[e] = 0.f;
                }
                *(uint2*)(st + p16 * (NC * 32) + j * 32 + q16 * 8) = make_uint2(Chunk<bf16>::pk(v[0], v[1]), Chunk<bf16>::pk(v[2], v[3]));
            }
            // the row back out: 16 pixels x co_fill channels, 16 bytes per lane and store, coalesced
            // (row base: uniform, on the scalar unit; the lane's piece: a 32-bit offset computed once per block)
            const long long rowpix = ((long long)(part * p.B + b) * p.h + y) * p.wd + tx0;
            unsigned char* orow = p.out + rowpix * p.ldo * 2;
            unsigned char* srow = p.sign_out ? p.sign_out + rowpix * (p.ldo >> 3) : nullptr;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NC / 2; ++k) {
                if (st_on[k]) {
                    const uint4 o = *(const uint4*)(st + (lane + 64 * k) * 16);
                    *(uint4*)(orow + st_off[k]) = o;
                    if (srow) srow[sg_off[k]] = (unsigned char)ups_sign_byte(o);
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
    // strips of 64 / 32 rows (16 / 8 per wave) where the height allows, 16-row tiles otherwise; the part-masked form (P rows out per row in)
    // keeps 32-row strips: 0.184 against 0.199 ms at 64 views x 10 parts (a quarter of the blocks, a longer tail)
    const int rw = (d->hi % 64 == 0 && k.P == 1) ? 16 : (d->hi % 32 == 0 ? 8 : 4);
    k.tiles_x = d->wi / 16; k.tiles_y = d->hi / (4 * rw);
    const long long blocks = (long long)k.B * k.tiles_x * k.tiles_y;
    if (blocks >= (1ll << 31) || (long long)d->hi * d->wi * d->ldi * 2 >= (1ll << 31)) return 1;      // (32-bit offsets inside an image)
#define UPS_FIRST_LAUNCH(NCV) do { \
        if (rw == 16) hipLaunchKernelGGL((conv3x3_first_kernel<NCV, 16>), dim3((unsigned)blocks), dim3(256), 0, s, k); \
        else if (rw == 8) hipLaunchKernelGGL((conv3x3_first_kernel<NCV, 8>), dim3((unsigned)blocks), dim3(256), 0, s, k); \
        else hipLaunchKernelGGL((conv3x3_first_kernel<NCV, 4>), dim3((unsigned)blocks), dim3(256), 0, s, k); } while (0)
    if (d->co_fill == 32) UPS_FIRST_LAUNCH(2); else UPS_FIRST_LAUNCH(4);
#undef UPS_FIRST_LAUNCH
    if (k.sign_out) g_ups_sign_written = 1;
    return 0;
}

