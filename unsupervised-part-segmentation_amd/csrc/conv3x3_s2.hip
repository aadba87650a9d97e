// Forward of the 3x3 / stride-2 `downsample` convolutions (cub/code/nn.py:816-817; five per encoder, one in the hourglass) at the
// resolutions where they are HBM streams: 32 / 64 input channels at 128x128 / 64x64 (encoder_1 on the P x B part images: 1.0 GB /
// 0.5 GB of tensor per launch).  gfx950 only, bf16.
//
// The generic gather kernel (conv_igemm.hip) ran them at 1.9 / 1.3 TB/s: nine separate tap gathers of every-other-pixel 64-byte
// segments into LDS, two barriers per chunk.  A stride-2 convolution uses every input pixel only 2.25 times (not 9), so -- unlike
// the stride-1 layers, where that experiment lost -- the taps can be read STRAIGHT FROM GLOBAL MEMORY as MFMA
// operands and the L1 / L2 carry the reuse (conv3x3_first.hip's scheme):
//   * v_mfma_f32_16x16x32_bf16, weights as the row operand: lane (p16, q16) of a step supplies the 16 bytes [8 q16, 8 q16 + 8) of
//     one 32-channel chunk of input pixel (2 y + r, 2 (x0 + p16) + s) -- a 16-byte global load, zero beyond the image (TF 'SAME'
//     pads after only for even sizes, Appendix A.1; general offsets come from the descriptor's taps);
//   * the block's weight set [9 taps][KC chunks][64 couts][64 B] sits in LDS (36 / 72 KB), written once with the conflict-free
//     XOR swizzle of the patch kernel's weight rows; B fragments are re-read per step (one ds_read_b128 per 2 MFMAs);
//   * a block walks a band of 32 output rows x 16 columns x 64 output channels with its weights resident (one weight load per 8
//     wave tasks); a wave task = 2 output rows: the 15 fragments (5 input rows x 3 column taps) of a 32-channel chunk are requested
//     at once and the next (task, chunk) is in flight while this one's 72 MFMAs run; no block barrier after the weights are in;
//   * epilogue: bias + CoordConv affine / class table (as conv_igemm), stored activation, bf16; each finished row goes through a
//     wave-private 2 KB LDS row and leaves in 16-byte coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/upsparts_hip.h"
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

struct S2K {
    const unsigned char* in; const unsigned char* w; unsigned char* out;
    const float* bias; const float* coord_tab;
    int n, hi, wi, ldi, ho, wo, co, co_fill, ldo, out_act, tiles_x, bands, band_rows, co_blocks;
    int dy[3], dx[3];                 // input offset of kernel row r / column s relative to (2 i, 2 j)
    float slope;
};

__device__ __forceinline__ int w_swz(int row) { return ((row >> 2) & 1) << 1; }     // (g, g^2, g, g^2): conv3x3_patch.hip a_swz16

template <int KC>       // 32-channel chunks of the input (1: 32 channels, 2: 64)
__global__ __launch_bounds__(256, 2) void conv3x3_s2_kernel(const S2K p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // weights [9][KC][64][64 B], then 4 x 2 KB row stages
    unsigned char* wl = smem;
    unsigned char* stage = smem + 9 * KC * 64 * 64 + (threadIdx.x >> 6) * 2048;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int p16 = lane & 15, q16 = lane >> 4;
    int t = blockIdx.x;
    const int cb = t % p.co_blocks; t /= p.co_blocks;          // the co-blocks of one tile run back to back (input shared in L2)
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int band = t % p.bands; const int img = t / p.bands;
    const int x0 = tx * 16;
    const int co0 = cb * 64;

    // ---- weights of this co-block into LDS, ONCE per block (a block then walks a band of output rows): global [tap][kc][co][64 B]
    // (ups_weight_prep, blocked-K) -> [tap*KC + kc][64 rows][64 B]
    for (int i = threadIdx.x; i < 9 * KC * 64 * 4; i += 256) {
        const int slot = i & 3, row = (i >> 2) & 63, tk = i >> 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (co0 + row < p.co) v = *(const uint4*)(p.w + (((long long)tk * p.co + co0 + row) * 4 + slot) * 16);
        *(uint4*)(wl + (tk * 64 + row) * 64 + ((slot ^ w_swz(row)) << 4)) = v;
    }
    __syncthreads();

    const unsigned char* inb = p.in + (long long)img * p.hi * p.wi * p.ldi * 2;
    // A task = two output rows (y, y + 1) x 16 columns of one 32-channel chunk: the FIVE input rows 2 y + dy[0] .. + 4 it touches
    // (output row i uses rows 2 i + r) x three column taps = 15 fragments, all requested at once (one global round trip per task and
    // chunk); the next (task, chunk) is in flight while this one's 72 MFMAs run.
    auto load_a = [&](int y, int kc, uint4 (&a)[5][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 5; ++rr)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ys = 2 * y + p.dy[0] + rr, xs = 2 * (x0 + p16) + p.dx[s];
                const bool ok = (unsigned)ys < (unsigned)p.hi && (unsigned)xs < (unsigned)p.wi;
                const long long off = ok ? ((long long)ys * p.wi + xs) * p.ldi * 2 + kc * 64 + q16 * 16 : 0;
                const uint4 v = *(const uint4*)(inb + off);
                const unsigned km = ok ? 0xffffffffu : 0u;      // (an AND, not a select: under the branch hipcc makes of it the load waits vmcnt(0))
                a[rr][s] = make_uint4(v.x & km, v.y & km, v.z & km, v.w & km);
            }
    };
    const float oact_ns = ups_slope_eff(p.out_act, p.slope);
    const float oact_e = p.out_act ? oact_ns : 1.f;        // max(f, 1 f) = f: no branch
    // the per-channel terms as 16-byte loads: whole groups of four channels, 16-byte aligned vectors
    const bool epi_vec = (p.co & 3) == 0 && ((((unsigned long long)p.bias) | ((unsigned long long)p.coord_tab)) & 15ull) == 0;
    const int x = x0 + p16;
    int xm = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) xm |= ((unsigned)(2 * x + p.dx[s]) < (unsigned)p.wi ? 1 : 0) << s;
    // the band's row pairs are dealt to the four waves round-robin: waves that run together read neighbouring input rows
    const int y_band = band * p.band_rows;
    const int ntask = p.band_rows / 8;                           // tasks per wave
    f32x4v acc[2][4];
    uint4 a_cur[5][3], a_nxt[5][3];
    load_a(y_band + wid * 2, 0, a_cur);
    for (int tk = 0; tk < ntask; ++tk) {
        const int y0 = y_band + (tk * 4 + wid) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (kc + 1 < KC) load_a(y0, kc + 1, a_nxt);
            else if (tk + 1 < ntask) load_a(y0 + 8, 0, a_nxt);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const unsigned char* wt = wl + ((3 * r + s) * KC + kc) * (64 * 64);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 16 * j + p16;
                        bf16x8 b = *(const bf16x8*)(wt + row * 64 + ((q16 ^ w_swz(row)) << 4));
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bf16x8 a;
                            __builtin_memcpy(&a, &a_cur[2 * i + r][s], 16);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[i][j], 0, 0, 0);
                        }
                    }
                }
            if (kc + 1 < KC || tk + 1 < ntask) {
#pragma unroll
                for (int rr = 0; rr < 5; ++rr)
#pragma unroll
                    for (int s = 0; s < 3; ++s) a_cur[rr][s] = a_nxt[rr][s];
            }
        }
        // ---- epilogue of the two rows.  lane (p16, q16) holds channels co0 + 16 j + 4 q16 + e of output pixel (y0 + i, x0 + p16)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = y0 + i;
            int ym = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) ym |= ((unsigned)(2 * y + p.dy[r]) < (unsigned)p.hi ? 1 : 0) << r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
                const int col4 = co0 + 16 * j + 4 * q16;
                if (epi_vec) {
                    // (round 6, late: the lane's four channels with ONE unconditional 16-byte load per term -- clamped channel, the weight
                    // tensor as a stand-in address for an absent bias / table -- instead of sixteen dword loads under branches per (row, j))
                    const bool ok4 = col4 < p.co;
                    const int colc = ok4 ? col4 : 0;
                    const float* bp = (p.bias ? p.bias : (const float*)p.w) + colc;
                    const float* tb = (p.coord_tab ? p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co : (const float*)p.w) + colc;
                    const int tstr = p.coord_tab ? p.co : 0;
                    const float4 b4 = *(const float4*)bp, t04 = *(const float4*)tb, t14 = *(const float4*)(tb + tstr), t24 = *(const float4*)(tb + 2 * tstr);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, a0[4] = {t04.x, t04.y, t04.z, t04.w};
                    const float a1[4] = {t14.x, t14.y, t14.z, t14.w}, a2[4] = {t24.x, t24.y, t24.z, t24.w};
                    const bool kb = ok4 && p.bias != nullptr, kt = ok4 && p.coord_tab != nullptr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float f = acc[i][j][e] + (kb ? bb[e] : 0.f);
                        if (kt) f += a0[e] + (float)x * a1[e] + (float)y * a2[e];
                        f = ups_vmax(f, oact_e * f);
                        v[e] = ok4 ? f : 0.f;
                    }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int col = co0 + 16 * j + 4 * q16 + e;
                    float f = acc[i][j][e];
                    if (col < p.co) {
                        if (p.bias) f += p.bias[col];
                        if (p.coord_tab) {
                            const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col;
                            f += tb[0] + (float)x * tb[p.co] + (float)y * tb[2 * p.co];
                        }
                        if (p.out_act) f = ups_vmax(f, oact_ns * f);
                    } else {
                        f = 0.f;
                    }
                    v[e] = f;
                }
                }
                *(uint2*)(stage + p16 * 128 + j * 32 + q16 * 8) = make_uint2(Chunk<bf16>::pk(v[0], v[1]), Chunk<bf16>::pk(v[2], v[3]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned char* orow = p.out + ((((long long)img * p.ho + y) * p.wo + x0) * p.ldo + co0) * 2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k;                     // 16-byte piece of the staged row: pixel idx / 8, piece idx % 8
                const int px = idx >> 3, ch = idx & 7;
                if (co0 + ch * 8 < p.co_fill) *(uint4*)(orow + (long long)px * p.ldo * 2 + ch * 16) = *(const uint4*)(stage + idx * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the staging row is rewritten by the wave's next row)
        }
    }
}

}  // namespace

// Internal entry of ups_conv_igemm's dispatcher.  Returns 1 if the problem is not one of these, 0 when launched, < 0 on a set-up error.
int ups_conv3x3_s2_try(const ups_conv_desc* d, hipStream_t s) {
    const char* env = getenv("UPS_S2_KERNEL");              // (read per call) "0": off (A/B runs); "force": also small launches (tests)
    if (env && env[0] == '0') return 1;
    const bool force = env && env[0] == 'f';
    if (d->dtype != UPS_BF16 || d->ntaps != 9 || d->kh != 3 || d->kw != 3 || d->in_sy != 2 || d->in_sx != 2 || d->out_sy != 1 ||
        d->out_sx != 1 || d->out_oy || d->out_ox || d->out_h != d->ho || d->out_w != d->wo)
        return 1;
    if (!(d->ci == 32 || d->ci == 64) || d->ldi != d->ci || (d->ho % 8) || (d->wo % 16) || d->act_in != UPS_ACT_NONE || d->res ||
        d->dact || d->d2s || d->out_f32 || d->mask_bits || d->mask_grad || d->f8_deq || d->in_f8 || d->out_f8 || d->out_f8_amax ||
        d->res_act || (d->ldo & 7) || (d->co_fill & 7) || d->co_fill > d->ldo)
        return 1;
    // only where the launch is a memory stream that fills the chip (small maps stay on the generic kernel's split-K forms)
    // (round 6, late: 128 Ki output pixels, was 256 Ki -- encoder_0's second downsample, 64 -> 128 channels at 64 x 64 x 128 images, runs in
    // 0.069 ms here against 0.095 on the generic kernel)
    if (!force && (long long)d->n * d->ho * d->wo < 128ll * 1024) return 1;
    // the kernel moves 16 bytes per lane (uint4 loads of the taps, uint4 stores of the staged row) and reads input row
    // 2 y + dy[0] + r for tap row r: both must hold, or the generic kernel takes the launch
    if ((((uintptr_t)d->in) | ((uintptr_t)d->out) | ((uintptr_t)d->w)) & 15) return 1;
    S2K k;
    for (int r = 0; r < 3; ++r) {
        if (d->tap_dy[3 * r] != d->tap_dy[0] + r || d->tap_dx[r] != d->tap_dx[0] + r) return 1;
        k.dy[r] = d->tap_dy[3 * r]; k.dx[r] = d->tap_dx[r];
        for (int c = 0; c < 3; ++c)
            if (d->tap_dy[3 * r + c] != d->tap_dy[3 * r] || d->tap_dx[3 * r + c] != d->tap_dx[c] || d->tap_w[3 * r + c] != 3 * r + c) return 1;
    }
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.out = (unsigned char*)d->out;
    k.bias = d->bias; k.coord_tab = d->coord_tab;
    k.n = d->n; k.hi = d->hi; k.wi = d->wi; k.ldi = d->ldi; k.ho = d->ho; k.wo = d->wo; k.co = d->co; k.co_fill = d->co_fill;
    k.ldo = d->ldo; k.out_act = d->out_act; k.slope = d->act_slope;
    // a block walks a band of output rows with its weights resident: 32 rows where the image allows (4 tasks per wave), else 16 / 8
    k.band_rows = (d->ho % 32 == 0) ? 32 : ((d->ho % 16 == 0) ? 16 : 8);
    k.tiles_x = d->wo / 16; k.bands = d->ho / k.band_rows; k.co_blocks = (d->co_fill + 63) / 64;
    const long long blocks = (long long)k.n * k.tiles_x * k.bands * k.co_blocks;
    if (blocks >= (1ll << 31)) return 1;
    const int KC = d->ci / 32;
    const size_t shm = (size_t)9 * KC * 64 * 64 + 4 * 2048;
    static UpsPerDevice a1, a2;
    if (KC == 1) {
        if (!a1) { if (hipFuncSetAttribute((const void*)conv3x3_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return UPS_E_LAUNCH; a1 = true; }
        hipLaunchKernelGGL((conv3x3_s2_kernel<1>), dim3((unsigned)blocks), dim3(256), shm, s, k);
    } else {
        if (!a2) { if (hipFuncSetAttribute((const void*)conv3x3_s2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return UPS_E_LAUNCH; a2 = true; }
        hipLaunchKernelGGL((conv3x3_s2_kernel<2>), dim3((unsigned)blocks), dim3(256), shm, s, k);
    }
    return 0;
}
