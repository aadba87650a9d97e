// Patch-tiled weight gradient for 3x3 / stride-1 'SAME' convolutions, bf16 (the wgrad twin of conv3x3_patch.hip):
//     dV[tap][ci][co] = sum_{img,y,x} act(in)[img][y+dy][x+dx][ci] * dout[img][y][x][co]      (nn.py:661-663)
//
// Block = 512 threads = 8 waves; it owns a (CB input channels x BN output channels) slice of dV for ALL nine taps
// (9 accumulator blocks of 32x32 per wave = 144 AGPRs) and walks over half-tiles of 8x16 output pixels:
// per half-tile the (8+2)x(16+2) input halo patch [pixel][CB] and the dout tile [pixel][BN] are staged once in LDS
// (row-major, double-buffered, activation-on-load applied once) and every tap's operand is a shifted window of the
// same patch, fetched as an MFMA fragment with the transposing read ds_read_b64_tr_b16 (pixels are the GEMM K).
// Staging traffic per MFMA is ~6x lower than in the generic split-K kernel (which re-stages X per tap).
// Waves are arranged WCI x WCO x WK: input-channel half, output-channel block, and K (tile rows) split; each K-part
// parts of a block are summed through LDS and each block writes one fp32 slab (deterministic reduce, no atomics).  The bias gradient is summed on the VALU from the dout
// fragments the channel-slice-0 waves already hold.
#include <stdlib.h>

#include "common.h"

namespace {

// A unit is TH rows x 16 pixels of one image: half-tiles (TH = 8; 18 x 10 = 180 patch pixels) for the wide variants, whose two
// LDS buffers would not fit otherwise, whole 16x16 tiles (TH = 16) for the thin ones (CB = 32, BN <= 64): those are bound by the
// load -> LDS -> barrier round trip of a unit, and a unit twice as large halves the number of round trips.
constexpr int TW = 16;
constexpr int PWID = TW + 2;

struct Wg3K {
    int n, h, w, ci, ldi, ci_log, cin_v, co, ldo, act_in, want_bias, units_total, units_per, tiles_x, tiles_y;
    float act_slope;
    unsigned long long tap_off, tap_wi;
    const void* in; const void* dout; float* ws;
    const unsigned* mask; int mask_B;     // part-masked input (ups_wgrad_desc.mask_*): in = view [mask_B,h,w,ldi], image = p*mask_B + b
    int in_f16;                           // `in` is an fp16 tensor (ups_wgrad_desc.in_f16): converted to bf16 while it is staged
    int taps_std;                         // the taps are in the forward's r-major order: dy = t/3 - 1, dx = t%3 - 1 (launcher)
};

__device__ inline int g_dy(unsigned long long off, int t) { return (int)((off >> (4 * t + 2)) & 3) - 1; }
__device__ inline int g_dx(unsigned long long off, int t) { return (int)((off >> (4 * t)) & 3) - 1; }
__device__ inline int g_w(unsigned long long wi, int t) { return (int)((wi >> (4 * t)) & 15); }

constexpr int lds_stride3(int row_bytes) { return ((row_bytes + 63) / 128) * 128 + 64; }

typedef __attribute__((address_space(3))) s16x4 lds3_s16x4;

// element j of lane l = M[row0 + 8*(l>>5) + j][c0 + (l&31)] of a row-major LDS tile (row stride rs bytes)
__device__ inline bf16x8 tr_frag3(const unsigned char* tile, int rs, int row0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, h = g >> 1;
    const int cb = c0 + 16 * (g & 1), q = i >> 2, pp = i & 3;
    const unsigned char* a0 = tile + (row0 + 8 * h + q) * rs + (cb + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds3_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds3_s16x4*)(a0 + 4 * rs));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

// D-tile swizzle: the dout tile is written by LDS-DMA, which fills 64 lanes x 16 B contiguously, so its rows are unpadded
// (BN * 2 bytes).  The transposing fragment read touches 4 consecutive rows x one 64-byte block per LDS cycle; the block
// index is XORed with a function of the row so that those four windows fall into different 64-byte bank quarters.
template <int BN> __device__ __forceinline__ int d_swz(int row) {
    if constexpr (BN == 128) return row & 3;
    else if constexpr (BN == 64) return (row >> 1) & 1;
    else return 0;
}

// dout fragment of tile row block row0 (multiple of 16), 32-channel block wco: as tr_frag3 on the swizzled unpadded tile
template <int BN>
__device__ inline bf16x8 tr_frag_d(const unsigned char* tile, int row0, int wco, int lane) {
    constexpr int RS = BN * 2;
    const int g = lane >> 4, i = lane & 15, h = g >> 1, q = i >> 2, pp = i & 3;
    const int row = row0 + 8 * h + q;
    const unsigned char* a0 = tile + row * RS + ((wco ^ d_swz<BN>(row)) << 6) + 32 * (g & 1) + 8 * pp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds3_s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds3_s16x4*)(a0 + 4 * RS));     // d_swz(row + 4) == d_swz(row)
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo; u.s[1] = hi;
    return u.b;
}

#if defined(UPS_WGRAD_NO_PIPE)
constexpr bool PIPE_X = false;
#else
constexpr bool PIPE_X = true;
#endif

template <int CB, int BN, int TH, bool SLIDE = false>
__global__ __launch_bounds__(512) void conv_wgrad3x3_kernel(const Wg3K p, const int cit, const int cot, const int nsplit) {
    constexpr int PROWS = TH + 2, PPIX = PWID * PROWS;
    constexpr int WCI = CB / 32, WCO = BN / 32, WK = 8 / (WCI * WCO);
    constexpr int RSX = lds_stride3(CB * 2), RSD = BN * 2;
    constexpr int XB = PPIX * RSX, DB = TH * TW * RSD;
    constexpr int CPX = CB / 8, CPD = BN / 8;                 // 16-byte chunks per row
    constexpr int NX = (PPIX * CPX + 511) / 512;              // X items per thread: 3 / 2
    constexpr int NWD = TH * TW * CPD / 512;                  // dout DMA wave-instructions per wave and unit: 4 / 2 / 1
    static_assert(NX <= 3 && NWD >= 1, "staging layout");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Xbuf = smem;                 // 2 x XB
    unsigned char* Dbuf = smem + 2 * XB;        // 2 x DB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float act_ns = ups_slope_eff(p.act_in, p.act_slope);   // branch-free activation-on-load
    // XCD-aware order: the cit * cot blocks that walk the SAME units (one K split) sit on one XCD, so the X / dout slices
    // they share are fetched from HBM once and served to the others by that XCD's L2 (blocks are dealt round-robin over
    // the 8 XCDs; dealt by (pair, split) every block of a split would land on a different one)
    const int pairs = cit * cot;
    int pair = blockIdx.x % pairs, split = blockIdx.x / pairs;
    if ((nsplit & 7) == 0) {
        const int j = blockIdx.x >> 3;
        pair = j % pairs;
        split = (j / pairs) * 8 + (blockIdx.x & 7);
    }
    const int cot_i = pair % cot, cit_i = pair / cot;
    const int w_ci = wid % WCI, w_co = (wid / WCI) % WCO, w_k = wid / (WCI * WCO);
    const int ci0 = cit_i * CB, co0 = cot_i * BN;
    const bf16* __restrict__ in = (const bf16*)p.in;
    const bf16* __restrict__ dout = (const bf16*)p.dout;
    const bool do_bias = p.want_bias && cit_i == 0 && w_ci == 0;

    const int u_begin = split * p.units_per;
    const int u_end = min(p.units_total, u_begin + p.units_per);

    // ---- staging.  Everything that depends only on the thread is decoded once; a unit contributes scalar bases.
    // X (input halo patch, activation-on-load): item k = tid + 512 k -> patch pixel (py, px), 16-byte chunk cc.
    //   xr[k]  pixel index relative to the patch origin in IMAGE pitch (py * w + px)
    //   xs[k]  LDS byte offset of the slot | edge flags << 16 (bit 0 top halo row, 1 bottom, 2 left column, 3 right,
    //          4 = never valid: past the item count or the channel count)
    unsigned xr[NX], xs[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int item = min(tid + 512 * k, PPIX * CPX - 1);
        const int pix = item / CPX, cc = item - pix * CPX;
        const int py = pix / PWID, px = pix - py * PWID;
        unsigned fl = (py == 0 ? 1u : 0u) | (py == PROWS - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == PWID - 1 ? 8u : 0u);
        if (tid + 512 * k >= PPIX * CPX || ci0 + cc * 8 >= p.ci) fl |= 16u;
        xr[k] = (unsigned)(py * p.w + px);
        xs[k] = (unsigned)(pix * RSX + cc * 16) | (fl << 16);
    }
    const unsigned x_rowb = (unsigned)p.ldi * 2u;                     // bytes per pixel of the input
    const unsigned x_chb = (unsigned)(ci0 + (tid % CPX) * 8) * 2u;    // 512 % CPX == 0: one chunk column per thread
    // dout tile by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write): wave-instruction j = wid + 8 q fills the LDS
    // slots j*64 + lane (16 B each) of the unpadded tile; slot (row, s) receives chunk s ^ (d_swz(row) << 2) (source-side
    // swizzle).  dd[q] = the lane's byte offset from the unit's first dout pixel.
    unsigned dd[NWD];
#pragma unroll
    for (int q = 0; q < NWD; ++q) {
        const int L = (wid + 8 * q) * 64 + lane, row = L / CPD, sl = L - row * CPD;
        const int c = sl ^ (d_swz<BN>(row) << 2);
        const int ch = min(co0 + c * 8, p.ldo - 8);                   // channels past ldo: a valid duplicate (never stored)
        dd[q] = (unsigned)((((row >> 4) * p.w + (row & 15)) * p.ldo + ch) * 2);
    }
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

    uint4 rx[NX];
    uint4 rxn[NX];            // pipelined SLIDE variant: the X items of the unit after next
    bool rxok[NX];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

    auto unit_origin = [&](int u, int& img, int& y0, int& x0) {
        const int half = TH == 8 ? (u & 1) : 0;
        int t = TH == 8 ? (u >> 1) : u;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        img = t / p.tiles_y;
        y0 = ty * 16 + half * TH; x0 = tx * TW;
    };
    // X loads of unit u into registers, UNCONDITIONALLY (lanes whose item lies outside the image read the tensor's first bytes and
    // drop them): the number of vector-memory operations per wave is then static, and a counted s_waitcnt can leave exactly
    // these loads in flight across a barrier (pipelined SLIDE variant below)
    auto load_x = [&](int u) __attribute__((always_inline)) {
        int img, y0, x0;
        unit_origin(u, img, y0, x0);
        const unsigned edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.h ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.w ? 8u : 0u) | 16u;
        const long long org = ((long long)img * p.h + (y0 - 1)) * p.w + (x0 - 1);
        const unsigned char* xb = (const unsigned char*)(in + org * p.ldi);
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            unsigned r = xr[k], f = xs[k];
            asm volatile("" : "+v"(r), "+v"(f));
            const bool ok = ((f >> 16) & edge) == 0u;
            const unsigned char* src = ok ? xb + (__umul24(r, x_rowb) + x_chb) : (const unsigned char*)in;
            uint4 v;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");
            rxn[k] = v;
            rxok[k] = ok;
        }
    };
    auto dma_d = [&](int u, int buf) __attribute__((always_inline)) {
        int img, y0, x0;
        unit_origin(u, img, y0, x0);
        const unsigned char* db = (const unsigned char*)(dout + (((long long)img * p.h + y0) * p.w + x0) * p.ldo);
#pragma unroll
        for (int q = 0; q < NWD; ++q) {
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(2 * XB + buf * DB + (wid + 8 * q) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lds_dst), "v"(dd[q]), "s"(db) : "memory", "m0");
        }
    };
    auto store_xn = [&](int buf) __attribute__((always_inline)) {
        unsigned char* X = Xbuf + buf * XB;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            unsigned f = xs[k];
            asm volatile("" : "+v"(f));
            uint4 v = rxok[k] ? rxn[k] : zero4;
            if (p.in_f16) v = ups_act_chunk_f16_to_bf16(v, act_ns, p.act_in != UPS_ACT_NONE);
            else if (p.act_in != UPS_ACT_NONE) v = ups_act_chunk(v, act_ns, (bf16*)nullptr);
            if (k + 1 < NX || tid + 512 * k < PPIX * CPX) *(uint4*)(X + (f & 0xffffu)) = v;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // global loads of unit u: X into registers, dout straight into D buffer `buf`
    auto load_unit = [&](int u, int buf) __attribute__((always_inline)) {
        int img, y0, x0;
        unit_origin(u, img, y0, x0);
        const unsigned edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.h ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.w ? 8u : 0u) | 16u;
        // part-masked input (ups_wgrad_desc.mask_*): image = part * B + b reads view image b where bit `part` is set
        const int img_x = p.mask ? img % p.mask_B : img;
        const int part = p.mask ? img / p.mask_B : 0;
        const long long org = ((long long)img_x * p.h + (y0 - 1)) * p.w + (x0 - 1);     // patch origin (may lie before the tensor)
        const unsigned char* xb = (const unsigned char*)(in + org * p.ldi);
        const unsigned* mb = p.mask ? p.mask + org : nullptr;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            unsigned r = xr[k], f = xs[k];
            asm volatile("" : "+v"(r), "+v"(f));      // per-unit address arithmetic stays inside the loop (no hoisted copies)
            uint4 v = zero4;
            if (((f >> 16) & edge) == 0u) {
                v = *(const uint4*)(xb + (__umul24(r, x_rowb) + x_chb));
                if (mb && !((mb[r] >> part) & 1u)) v = zero4;
            }
            rx[k] = v;
        }
        const unsigned char* db = (const unsigned char*)(dout + (((long long)img * p.h + y0) * p.w + x0) * p.ldo);
#pragma unroll
        for (int q = 0; q < NWD; ++q) {
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(2 * XB + buf * DB + (wid + 8 * q) * 1024));
            // inline asm: see conv3x3_patch.hip (the builtin makes hipcc wait lgkmcnt(0) before every later LDS use)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lds_dst), "v"(dd[q]), "s"(db) : "memory", "m0");
        }
    };
    auto store_unit = [&](int buf) __attribute__((always_inline)) {
        unsigned char* X = Xbuf + buf * XB;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            unsigned f = xs[k];
            asm volatile("" : "+v"(f));
            uint4 v = rx[k];
            if (p.in_f16) v = ups_act_chunk_f16_to_bf16(v, act_ns, p.act_in != UPS_ACT_NONE);
            else if (p.act_in != UPS_ACT_NONE) v = ups_act_chunk(v, act_ns, (bf16*)nullptr);
            if (k + 1 < NX || tid + 512 * k < PPIX * CPX) *(uint4*)(X + (f & 0xffffu)) = v;
            __builtin_amdgcn_sched_barrier(0);        // one item at a time (register pressure)
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;        // bias gradient: column sums straight from the dout fragments (k = 8*(lane>>5)+j, col = lane&31)

    if constexpr (WK == 1 && SLIDE) {
      if (!p.mask && PIPE_X) {
        // Software-pipelined staging: the X items of unit u+1 are converted and written to LDS in the MIDDLE of unit u's MFMAs
        // (they were requested during unit u-1), the loads of unit u+2 are issued right behind and stay in flight across the
        // barrier (counted wait: only the dout DMA of unit u+1, issued first, has to have landed).  One block per CU runs its
        // eight waves in lock-step between barriers, so a load -> wait -> convert -> ds_write phase at the end of every unit
        // left the matrix pipe idle for its whole length.
        static_assert(NX == 3 && (NWD == 4 || NWD == 2 || NWD == 1), "pipelined staging: item / DMA piece counts");
#define UPS_WAIT_X(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(rxn[0].x), "+v"(rxn[0].y), "+v"(rxn[0].z), "+v"(rxn[0].w), \
                                   "+v"(rxn[1].x), "+v"(rxn[1].y), "+v"(rxn[1].z), "+v"(rxn[1].w), "+v"(rxn[2].x), "+v"(rxn[2].y), \
                                   "+v"(rxn[2].z), "+v"(rxn[2].w) :: "memory")
        if (u_begin < u_end) {
            load_x(u_begin); dma_d(u_begin, 0);
            UPS_WAIT_X(0);
            store_xn(0);
            if (u_begin + 1 < u_end) load_x(u_begin + 1);
        }
        UPS_WAIT_X(0);
        __syncthreads();
        for (int u = u_begin; u < u_end; ++u) {
            const int buf = (u - u_begin) & 1;
            if (u + 1 < u_end) dma_d(u + 1, buf ^ 1);
            const unsigned char* X = Xbuf + buf * XB;
            const unsigned char* D = Dbuf + buf * DB;
            bf16x8 aw[3][3];
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                aw[0][dxi] = tr_frag3(X, RSX, dxi, w_ci * 32, lane);
                aw[1][dxi] = tr_frag3(X, RSX, PWID + dxi, w_ci * 32, lane);
            }
            bf16x8 bcur = tr_frag_d<BN>(D, 0, w_co, lane);
#pragma unroll
            for (int ks = 0; ks < TH; ++ks) {
                if (ks == TH / 2) {
                    if (u + 1 < u_end) {
                        // the X items of unit u+1 (requested half a unit ago or earlier) are older than this unit's NWD dout DMA
                        // pieces; the registers are operands of the wait so that no use of them is scheduled above it (the
                        // loads are inline asm: the compiler's own wait-count bookkeeping does not see them)
                        if (NWD == 4) UPS_WAIT_X(4); else if (NWD == 2) UPS_WAIT_X(2); else UPS_WAIT_X(1);
                        store_xn(buf ^ 1);
                    }
                    if (u + 2 < u_end) load_x(u + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi) aw[(ks + 2) % 3][dxi] = tr_frag3(X, RSX, (ks + 2) * PWID + dxi, w_ci * 32, lane);
                // the dout fragment of the NEXT row is requested now: all nine MFMAs of a row wait for theirs
                const bf16x8 bnext = tr_frag_d<BN>(D, (ks + 1 < TH ? ks + 1 : ks) * TW, w_co, lane);
                const bf16x8 b = bcur;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[(ks + t / 3) % 3][t % 3], b, acc[t], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += (float)b[j];
                }
                bcur = bnext;
                __builtin_amdgcn_sched_barrier(0);
            }
            // the dout DMA of unit u+1 (older than the NX loads of unit u+2) must have landed; those loads stay in flight
            if (u + 2 < u_end) {
                if (NX == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        goto reduce_parts;
      }
    }
    if (u_begin < u_end) { load_unit(u_begin, 0); store_unit(0); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        const int buf = (u - u_begin) & 1;
        if (u + 1 < u_end) load_unit(u + 1, buf ^ 1);
        const unsigned char* X = Xbuf + buf * XB;
        const unsigned char* D = Dbuf + buf * DB;
        if constexpr (WK == 1 && SLIDE) {
            // Sliding window over the tile rows (a wave that walks EVERY row of the unit, taps in r-major order): the operand of
            // tap (dy, dx) at row ks is the patch window of patch row ks + dy + 1 shifted by dx -- the same fragment serves
            // dy = +1 at row ks, dy = 0 at row ks + 1 and dy = -1 at row ks + 2.  Three patch rows x three shifts stay in
            // registers, every row brings in 3 new fragments + 1 dout fragment: 4 transposing fragment reads per 9 MFMAs
            // instead of 10 (the LDS reads were what bounded this kernel: 2.19 ms with the staging compiled out).
            bf16x8 aw[3][3];
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                aw[0][dxi] = tr_frag3(X, RSX, dxi, w_ci * 32, lane);
                aw[1][dxi] = tr_frag3(X, RSX, PWID + dxi, w_ci * 32, lane);
            }
#pragma unroll
            for (int ks = 0; ks < TH; ++ks) {
#pragma unroll
                for (int dxi = 0; dxi < 3; ++dxi) aw[(ks + 2) % 3][dxi] = tr_frag3(X, RSX, (ks + 2) * PWID + dxi, w_ci * 32, lane);
                const bf16x8 b = tr_frag_d<BN>(D, ks * TW, w_co, lane);
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[(ks + t / 3) % 3][t % 3], b, acc[t], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum += (float)b[j];
                }
                __builtin_amdgcn_sched_barrier(0);        // rows stay in program order: the scheduler otherwise pulls every
            }                                             // row's fragment reads to the front (36 -> 100+ live registers)
        } else
#pragma unroll 1
        for (int kk = 0; kk < TH / WK; ++kk) {
            const int ks = w_k + kk * WK;                               // tile row handled by this wave
            const bf16x8 b = tr_frag_d<BN>(D, ks * TW, w_co, lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // SLIDE instances are only launched for the r-major tap order: the window offsets are then immediates
                const int row0 = SLIDE ? (ks + t / 3) * PWID + t % 3 : (ks + g_dy(p.tap_off, t) + 1) * PWID + g_dx(p.tap_off, t) + 1;
                const bf16x8 a = tr_frag3(X, RSX, row0, w_ci * 32, lane);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)b[j];
            }
        }
        if (u + 1 < u_end) store_unit(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the DMA of the next dout tile has landed
        __syncthreads();
    }

reduce_parts:
    // ---- reduce the WK K-parts of the block through LDS (tap by tap), then the w_k == 0 waves write the block's slab
    bsum += __shfl_xor(bsum, 32, 64);                                  // the two lane halves own k = 0..7 / 8..15
    if (WK > 1) {
        float* red = (float*)smem;                                     // (WK-1) x (WCI*WCO) x 32x32 floats <= 28 KB
        const int slot = ((w_k - 1) * (WCI * WCO) + w_ci + WCI * w_co) * 1024;
        const int slot0 = (w_ci + WCI * w_co) * 1024;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            if (w_k > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[slot + e * 64 + lane] = acc[t][e];
            }
            __syncthreads();
            if (w_k == 0) {
                for (int k = 1; k < WK; ++k)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[t][e] += red[(k - 1) * (WCI * WCO) * 1024 + slot0 + e * 64 + lane];
            }
        }
        __syncthreads();
        if (w_k > 0) red[slot + lane] = bsum;
        __syncthreads();
        if (w_k == 0)
            for (int k = 1; k < WK; ++k) bsum += red[(k - 1) * (WCI * WCO) * 1024 + slot0 + lane];
    }
    const long long slab_sz = (long long)9 * p.cin_v * p.co + p.co;
    float* slab = p.ws + (long long)split * slab_sz;
    const int col = co0 + w_co * 32 + (lane & 31);
    if (w_k == 0 && col < p.co) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tw = g_w(p.tap_wi, t);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ci0 + w_ci * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < p.ci_log) slab[((long long)tw * p.cin_v + row) * p.co + col] = acc[t][e];
            }
        }
        if (do_bias && (lane >> 5) == 0) slab[(long long)9 * p.cin_v * p.co + col] = bsum;
    }
}

struct Variant { int cb, bn, wk, th; };

Variant pick(const ups_wgrad_desc* d) {
    Variant v;
    v.cb = d->ci > 32 ? 64 : 32;
    v.bn = d->co > 64 ? 128 : (d->co > 32 ? 64 : 32);
    v.wk = 8 / ((v.cb / 32) * (v.bn / 32));
    v.th = (v.cb == 32 && v.bn <= 64) ? 16 : 8;
    return v;
}

bool eligible(const ups_wgrad_desc* d) {
    if (d->dtype != UPS_BF16 || d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1) return false;
    if (d->hi != d->ho || d->wi != d->wo || d->hi % 16 || d->wi % 16) return false;
    // 32-bit byte offsets of a staged item from the unit's scalar base (patch of 10 rows; 24-bit operands of the v_mad_u32_u24)
    if (19ll * d->wi * d->ldi * 2 >= (1ll << 31) || 17ll * d->wi * d->ldo * 2 >= (1ll << 31)) return false;
    if (18ll * d->wi >= (1 << 24) || (long long)d->ldi * 2 >= (1 << 24)) return false;
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        const int dy = d->tap_dy[t], dx = d->tap_dx[t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || d->tap_w[t] < 0 || d->tap_w[t] > 8) return false;
        seen[(dy + 1) * 3 + dx + 1] = true;
    }
    for (int t = 0; t < 9; ++t) if (!seen[t]) return false;
    const char* force = getenv("UPS_FORCE_GENERIC_CONV");
    return !(force && force[0] == '1');
}

template <int CB, int BN, int TH, bool SLIDE = false>
int launch3(const Wg3K& k, int cit, int cot, int splitk, hipStream_t s) {
    constexpr int RSX = lds_stride3(CB * 2), RSD = BN * 2;
    constexpr int PPIX = PWID * (TH + 2);
    size_t shmem = 2 * (size_t)(PPIX * RSX + TH * TW * RSD);
    constexpr size_t red = (size_t)(8 / ((CB / 32) * (BN / 32)) - 1) * (CB / 32) * (BN / 32) * 4096;   // K-part reduction scratch
    if (shmem < red) shmem = red;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv_wgrad3x3_kernel<CB, BN, TH, SLIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shmem) != hipSuccess) return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_wgrad3x3_kernel<CB, BN, TH, SLIDE>), dim3(cit * cot * splitk), dim3(512), shmem, s, k, cit, cot, splitk);
    return UPS_OK;
}

}  // namespace

// Returns 1 when the problem is not eligible; otherwise fills block split count and slabs (= splitk * WK).
int ups_wgrad3x3_plan(const ups_wgrad_desc* d, int* splitk, int* slabs) {
    if (!eligible(d)) return 1;
    const Variant v = pick(d);
    const int pairs = ups_cdiv(d->ci, v.cb) * ups_cdiv(d->co, v.bn);
    const int units = d->n * (d->hi / 16) * (d->wi / 16) * (16 / v.th);
    // one block per CU for the big variant (151 KB of LDS per block): fewer, longer blocks halve the slab traffic
    int sk = ups_cdiv((v.cb == 64 && v.bn == 128) ? 256 : 512, pairs);
    if (sk > units / 4) sk = units / 4 > 0 ? units / 4 : 1;
    if (sk > 256) sk = 256;
    if (sk >= 8) sk &= ~7;                          // whole splits per XCD (see the block order in the kernel)
    *splitk = sk;
    *slabs = sk;                                   // the K-parts of a block are reduced in LDS
    return 0;
}

int ups_wgrad3x3_run(const ups_wgrad_desc* d, hipStream_t s) {
    const Variant v = pick(d);
    Wg3K k;
    k.n = d->n; k.h = d->hi; k.w = d->wi; k.ci = d->ci; k.ldi = d->ldi; k.ci_log = d->ci_log; k.cin_v = d->cin_v;
    k.co = d->co; k.ldo = d->ldo; k.act_in = d->act_in; k.act_slope = d->act_slope; k.want_bias = d->grad_bias != nullptr;
    k.tiles_x = d->wi / 16; k.tiles_y = d->hi / 16;
    k.units_total = d->n * k.tiles_x * k.tiles_y * (16 / v.th);
    k.units_per = ups_cdiv(k.units_total, d->splitk);
    k.in = d->in; k.dout = d->dout; k.ws = d->workspace;
    k.mask = d->mask_bits; k.mask_B = d->mask_batch;
    k.in_f16 = d->in_f16;
    k.taps_std = 1;
    for (int t = 0; t < 9; ++t) if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) k.taps_std = 0;
    { const char* e = getenv("UPS_WGRAD_SLIDE"); if (e && e[0] == '0') k.taps_std = 0; }      // A/B switch
    k.tap_off = 0; k.tap_wi = 0;
    for (int t = 0; t < 9; ++t) {
        k.tap_off |= (unsigned long long)(((d->tap_dy[t] + 1) << 2) | (d->tap_dx[t] + 1)) << (4 * t);
        k.tap_wi |= (unsigned long long)d->tap_w[t] << (4 * t);
    }
    const int cit = ups_cdiv(d->ci, v.cb), cot = ups_cdiv(d->co, v.bn);
    if (v.cb == 64 && v.bn == 128) return k.taps_std ? launch3<64, 128, 8, true>(k, cit, cot, d->splitk, s) : launch3<64, 128, 8>(k, cit, cot, d->splitk, s);
    // (SLIDE on the variants whose waves do not walk every row = the r-major tap order known at compile time, nothing else)
    if (v.cb == 64 && v.bn == 64) return k.taps_std ? launch3<64, 64, 8, true>(k, cit, cot, d->splitk, s) : launch3<64, 64, 8>(k, cit, cot, d->splitk, s);
    if (v.cb == 64 && v.bn == 32) return k.taps_std ? launch3<64, 32, 8, true>(k, cit, cot, d->splitk, s) : launch3<64, 32, 8>(k, cit, cot, d->splitk, s);
    if (v.cb == 32 && v.bn == 128) return k.taps_std ? launch3<32, 128, 8, true>(k, cit, cot, d->splitk, s) : launch3<32, 128, 8>(k, cit, cot, d->splitk, s);
    if (v.cb == 32 && v.bn == 64) return k.taps_std ? launch3<32, 64, 16, true>(k, cit, cot, d->splitk, s) : launch3<32, 64, 16>(k, cit, cot, d->splitk, s);
    return k.taps_std ? launch3<32, 32, 16, true>(k, cit, cot, d->splitk, s) : launch3<32, 32, 16>(k, cit, cot, d->splitk, s);
}
