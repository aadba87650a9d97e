// Row-streaming 3x3 / stride-1 'SAME' convolution for the THIN residual blocks of the encoders on large image batches
// (cub/code/nn.py:1042-1056 at 32 / 64 channels: encoder_1 runs them on the P x B part images, 1.3 GB of tensor per launch for
// 0.19 TFLOP -- HBM streams).  gfx950 only, 16-bit tensors.
//
// The patch-tiled kernel (conv3x3_patch.hip) runs these layers at 2.5 TB/s: a 16x16 tile is one block with its own prologue, its
// own copy of the 3x3 weight set through LDS, one dependent HBM round trip for its halo patch and an epilogue -- 9.5 us of block
// life for 0.5 us of MFMA work, 470 vector instructions per wave (tools/probes/phase_timing.py).  Here a block owns a band of 32
// full-width image rows and streams them:
//   * every input row is fetched ONCE, whole, by LDS-DMA (one 1 KiB global_load_lds_dwordx4 per wave and row) into a ring of NR row
//     buffers [32-channel plane][pixel slot = x + 1][64 B] (slots 0 and W + 1 stay zero: the horizontal padding); the swizzle of the
//     patch kernel is applied on the source side; rows are requested four (or eight) ahead of their use and waited for with a
//     counted s_waitcnt -- the only block barrier is one per two output rows;
//   * the wave's weights stay in REGISTERS for the whole band (9 taps x CI / 32 chunks x two 16-channel blocks: 72 / 144 VGPRs):
//     no weight traffic and no B-fragment reads inside the loop;
//   * a wave owns 16 columns x 32 output channels; per iteration it computes two output rows from four input rows: 12 A-fragment
//     reads (ds_read_b128, conflict-free) for 36 MFMAs per 32-channel chunk (v_mfma_f32_16x16x32, weights as the row operand);
//   * epilogue per row: bias, the residual from the resident centre row (res == in, stored post-activation: inverted on the way),
//     the stored activation, 16-bit pack, a wave-private 1.25 KB LDS stage, one 1 KiB coalesced store per wave and row.
// Input gradient of the same layers (DG): the operand is the gradient tensor, the taps are flipped, and the epilogue multiplies
// by act'(x) -- the sign of the stored forward input, fetched row by row into a second small ring by the same DMA -- before the
// residual gradient is added.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/upsparts_hip.h"
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(3))) float f32x3v;
template <typename T> struct RFrag;
template <> struct RFrag<bf16> { typedef bf16x8 type; };
template <> struct RFrag<f16> { typedef f16x8 type; };

struct RowsK {
    const unsigned char* in; const unsigned char* w; unsigned char* out;
    const float* bias; const unsigned char* dact;
    const unsigned char* dact_bits;     // ups_conv_desc.dact_bits: the sign bytes of the tensor `dact` points to (DG == 3) or NULL
    unsigned char* sign_out;            // ups_conv_desc.sign_out: one sign byte beside every stored 16-byte chunk, or NULL
    int n, h, ldi, ldo, ldd, co_tot, band_rows, bands, res_self, res_act, out_act;
    float slope, dact_ns;
};

// a row of zeros in global memory: input rows above / below the image are "fetched" from here, so that the loop has no row tests and
// the number of DMAs per iteration is static (W * ldi * 2 <= 256 KiB is checked by the launchers)
__device__ unsigned char ups_rows_zero[262144];

__device__ __forceinline__ int r_swz(int P) { return ((P >> 2) & 1) << 1; }     // (g, g^2, g, g^2): conv3x3_patch.hip a_swz16

// ---- Correctness rules of this file (round 6; the hunt and its evidence: docs/design/rows_hazard.md) ------------------------------
// Round 5 left conv3x3_rows_kernel<bf16,64,6,10,false,0> "correct by timing": one more never-taken conditional store in its epilogue
// (-DUPS_ROWS_FWD_SIGN, kept below as the reproducer) made output channel 12 of the even rows wrong in 99 launches of 100.  Round 6
// traced the wrong element, with listing-level patches that change ONE thing at a time (tools/asm_patch_build.sh), to the epilogue's
// first PACKED fp32 instruction, `v_pk_add_f32 acc, acc, residual op_sel:[0,1] op_sel_hi:[1,0]`: in lanes 48..63 its low half came back
// as `acc + 0` although both operands were read intact by the instructions right before and right after it -- only in the four waves
// that reach their epilogue while their SIMD sibling is still inside its MFMA section, only within ~16 cycles of a fixed point of the
// epilogue.  It is NOT an MFMA operand hazard (renaming every register involved off the MFMA operands changes nothing), not the row
// DMA, not an early LDS return.  Two independent cures, both measured at 0 wrong launches of 500 on the reproducer build and at no
// cost on the step (profiles/round6_rows_hazard_*.txt):
//   1. no packed fp32 VALU instructions in this translation unit: it is compiled with `-target-feature -packed-fp32-ops`
//      (csrc/flags.sh), and tools/check_listing.py fails the build if a v_pk_*_f32 appears in its listing;
//   2. ups_rows_fence(): 32 idle issue slots between the MFMA section and the epilogue of every iteration, pinned by scheduling
//      barriers, in every kernel of this file that runs its epilogue beside a sibling's MFMA section.
// The forms whose operands arrived by inline-asm REGISTER loads with hand-counted waits (hipcc is free to move such registers before
// the wait: the two-tile input gradient and the DG == 2 one-tile form) are deleted; the one kernel that still loads registers that way
// (conv3x3_rows_maskgrad_kernel) is checked by tools/check_asm_loads.py at build time.
__device__ __forceinline__ void ups_rows_fence() {
#ifndef UPS_ROWS_NO_FENCE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// CI input channels (32 / 64), image width 1 << LW (128 / 64): 8 waves = (W / 16 column tiles) x (8 / (W / 16) groups of 32 outputs),
// i.e. 32 outputs at W = 128, 64 at W = 64.  NR ring rows.  FLIP: the input gradient's tap order (dy = 1 - t / 3, dx = 1 - t % 3).
// DG: 0 forward-type epilogue; 1 act' from a second DMA ring of the forward input's rows (one block per CU at 32 channels);
// 3 (round 5): act' from the producer's SIGN BYTES (RowsK.dact_bits), the same second ring with 4 bytes per pixel and plane instead
// of 64 -- by LDS-DMA like everything else these kernels fetch inside their loop (no register results for hipcc to move).
// (DG == 2, act' by counted inline-asm register loads, was deleted in round 6: see the rules above.)
template <typename T, int CI, int LW, int NR, bool FLIP, int DG>
__global__ __launch_bounds__(512, (CI == 32 && DG == 0 && NR <= 8) ? 4 : 2) void conv3x3_rows_kernel(const RowsK p) {
    static_assert(DG == 0 || DG == 1 || DG == 3, "act' operand: none, the forward input's rows, or its sign bytes");
    constexpr int W = 1 << LW, KC = CI / 32, NCT = W / 16;
    constexpr int PL = (W + 2) * 64;            // bytes of one 32-channel plane of a row
    constexpr int RB = KC * PL;                 // bytes of a row buffer
    constexpr int ST = 16 * 80;                 // wave-private output stage: 16 pixels x (64 + 16) bytes
    constexpr int L = (NR - 4) / 2;             // iterations of lead of the row requests (NR = 2 L + 4)
    constexpr int DR = (DG == 1 || DG == 3) ? 2 * L + 2 : 0;   // ring rows of the act' operand (W pixels x 64 B x output-channel planes; no halo)
    constexpr int GS = (DG == 1 || DG == 3) ? 4 : 2;            // DMA instructions a wave issues per iteration
    constexpr int NCG = 8 / NCT;                // output-channel groups of 32 = planes of the output / act' rows
    constexpr int DRB = DG == 3 ? 8 * 256 : NCG * W * 64;       // (DG == 3: one dword per lane and wave: the pixel's four sign bytes of the wave's plane)
    static_assert(KC * NCT == 8 && NCG * NCT == 8, "one DMA piece per wave and row");
    typedef typename RFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                              // NR x RB
    unsigned char* dring = smem + NR * RB;                   // DR x DRB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem + NR * RB + DR * DRB + wid * ST;
    float* biasL = (float*)(smem + NR * RB + DR * DRB + 8 * ST);          // 64 floats
    const int p16 = lane & 15, q16 = lane >> 4;
    const int ct = wid % NCT, cg = wid / NCT;                // column tile, output-channel group (also: DMA segment, plane)
    const int band = blockIdx.x % p.bands, img = blockIdx.x / p.bands;
    const int y0 = band * p.band_rows;
    const int y1 = min(p.h, y0 + p.band_rows);

    // the halo slots of every ring row are zero for the whole kernel (the DMA never writes them)
    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);

    // ---- weights into registers: fragment (tap t, chunk kc, block j) = rows cg * 32 + 16 j + p16 of slice t, 16-byte piece q16
    frag_t wb[9][KC][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[t][kc][j] = *(const frag_t*)(p.w + ((long long)(t * KC + kc) * p.co_tot + cg * 32 + 16 * j + p16) * 64 + q16 * 16);
    if (tid < 64) biasL[tid] = (p.bias && tid < p.co_tot) ? p.bias[tid] : 0.f;
    __syncthreads();

    // ---- row DMA: wave (segment ct, plane cg) moves 16 pixels x 64 B of input row k (k = y - (y0 - 1)) into ring slot k % NR
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int Pd = 1 + 16 * ct + (lane >> 2);
    const unsigned d_off = (unsigned)((16 * ct + (lane >> 2)) * p.ldi * 2 + cg * 64 + (((lane & 3) ^ r_swz(Pd)) << 4));
    const unsigned char* in_img = p.in + (long long)img * p.h * W * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(W * p.ldi * 2);
    // act' operand (DG): wave (segment ct, plane cg) moves 16 pixels x 64 B of row y of the forward input (no halo, no swizzle
    // needed: the epilogue reads 8 bytes per lane at a 64-byte pixel pitch -- 2-way conflicts on a 4-instruction read)
    const unsigned dd_off = (unsigned)((16 * ct + (lane >> 2)) * p.ldd * 2 + cg * 64 + ((lane & 3) << 4));
    const unsigned char* da_img = DG == 1 ? p.dact + (long long)img * p.h * W * p.ldd * 2 : nullptr;
    auto issue_row = [&](int k) __attribute__((always_inline)) {
        const int y = y0 - 1 + k;
        const unsigned char* src = (unsigned)y < (unsigned)p.h ? in_img + (long long)y * row_bytes : ups_rows_zero;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + cg * PL + (1 + 16 * ct) * 64));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(d_off), "s"(src) : "memory", "m0");
    };
    const unsigned db_off = DG == 3 ? (unsigned)((16 * ct + (lane & 15)) * (p.ldd >> 3) + cg * 4) : 0u;
    const unsigned char* db_img = DG == 3 ? p.dact_bits + (long long)img * p.h * W * (p.ldd >> 3) : nullptr;
    auto issue_drow = [&](int ko) __attribute__((always_inline)) {          // ko = output row index inside the band
        if constexpr (DG == 3) {
            // lane l fetches the dword of pixel 16 ct + (l & 15) (lanes 16 .. 63 repeat the first sixteen: no exec games) into the
            // wave's 256-byte slot of act' ring row ko
            const int y = min(y0 + ko, p.h - 1);
            const unsigned char* src = db_img + (long long)y * (unsigned)(W * (p.ldd >> 3));
            const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(NR * RB + (ko % DR) * DRB + wid * 256));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(dst), "v"(db_off), "s"(src) : "memory", "m0");
        }
        if constexpr (DG == 1) {
            const int y = min(y0 + ko, p.h - 1);
            const unsigned char* src = da_img + (long long)y * (unsigned)(W * p.ldd * 2);
            const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(NR * RB + (ko % DR) * DRB + cg * (W * 64) + 16 * ct * 64));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(dd_off), "s"(src) : "memory", "m0");
        }
    };
    // prologue: what iterations -L .. -1 would have requested (iteration it requests the rows iteration it + L ends on)
    issue_row(0); issue_row(1);
#pragma unroll
    for (int i = 0; i < L; ++i) { issue_row(2 * i + 2); issue_row(2 * i + 3); issue_drow(2 * i); issue_drow(2 * i + 1); }

    const float oact_ns = ups_slope_eff(p.out_act, p.slope);
    const float res_inv = p.res_act ? 1.f / p.slope : 1.f;
    const int a_lane = (16 * ct + p16) * 64;            // + dx * 64 + swizzled piece
    const int iters = (y1 - y0 + 1) >> 1;
    unsigned char* out_img = p.out + (long long)img * p.h * W * p.ldo * 2;

    for (int it = 0; it < iters; ++it) {
        // The requests of iteration it - L (input rows up to 2 it + 3, act' rows 2 it, 2 it + 1) must have landed.  Younger operations
        // that may stay in flight: the requests of iterations it - L + 1 .. it - 1 (GS each) and, from the first real iteration on,
        // their two stores each (counted once: a lower bound that holds for every it >= 1)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * GS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * GS + 2) : "memory");
        __builtin_amdgcn_s_barrier();
        // the ring slots of input rows 2 it - 2, 2 it - 1 (and of the act' rows of iteration it - 1) are free now
        issue_row(2 * it + 2 * L + 2);
        issue_row(2 * it + 2 * L + 3);
        issue_drow(2 * (it + L)); issue_drow(2 * (it + L) + 1);

        f32x4v acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4v b4 = *(const f32x4v*)(biasL + cg * 32 + 16 * j + 4 * q16);
            acc[0][j] = b4; acc[1][j] = b4;
        }
        const int yb = y0 + 2 * it;                       // first output row of the iteration; input row of ring index r4 = yb - 1 + r4
        const unsigned char* rowp[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) rowp[r4] = ring + ((2 * it + r4) % NR) * RB;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int P = 16 * ct + p16 + dx;
            const int aoff = a_lane + dx * 64 + ((q16 ^ r_swz(P)) << 4);
            frag_t a[4][KC];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) a[r4][kc] = *(const frag_t*)(rowp[r4] + kc * PL + aoff);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int dyi = 0; dyi < 3; ++dyi) {
                    const int t = FLIP ? (2 - dyi) * 3 + (2 - dx) : dyi * 3 + dx;
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if constexpr (__is_same(T, bf16))
                                acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[t][kc][j], a[r + dyi][kc], acc[r][j], 0, 0, 0);
                            else
                                acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][kc][j], a[r + dyi][kc], acc[r][j], 0, 0, 0);
                        }
                }
        }
        ups_rows_fence();
        // ---- epilogue: lane (p16, q16) holds channels cg * 32 + 16 j + 4 q16 + e of pixel (row, 16 ct + p16)
        const int Pc = 16 * ct + p16 + 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = yb + r;
            if (y >= y1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[r][j][e];
                if constexpr (DG == 3) {
                    const unsigned w4 = *(const unsigned*)(dring + ((2 * it + r) % DR) * DRB + wid * 256 + p16 * 4);
                    const unsigned nib = w4 >> (8 * (2 * j + (q16 >> 1)) + 4 * (q16 & 1));      // the lane's four channels
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= (nib >> e) & 1u ? 1.f : p.dact_ns;
                }
                if constexpr (DG == 1) {
                    if (p.dact) {
                        const uint2 dv = *(const uint2*)(dring + ((2 * it + r) % DR) * DRB + cg * (W * 64) + (16 * ct + p16) * 64 + (16 * j + 4 * q16) * 2);
                        float d0, d1, d2, d3;
                        ups_unpack2<T>(dv.x, d0, d1); ups_unpack2<T>(dv.y, d2, d3);
                        v[0] *= d0 > 0.f ? 1.f : p.dact_ns; v[1] *= d1 > 0.f ? 1.f : p.dact_ns;
                        v[2] *= d2 > 0.f ? 1.f : p.dact_ns; v[3] *= d3 > 0.f ? 1.f : p.dact_ns;
                    }
                }
                if (p.res_self) {
                    const int c32 = 16 * j + 4 * q16;
                    const uint2 rr = *(const uint2*)(rowp[1 + r] + cg * PL + Pc * 64 + (((c32 >> 3) ^ r_swz(Pc)) << 4) + (q16 & 1) * 8);
                    float r0, r1, r2, r3;
                    ups_unpack2<T>(rr.x, r0, r1); ups_unpack2<T>(rr.y, r2, r3);
                    if (p.res_act) {
                        r0 = r0 > 0.f ? r0 : r0 * res_inv; r1 = r1 > 0.f ? r1 : r1 * res_inv;
                        r2 = r2 > 0.f ? r2 : r2 * res_inv; r3 = r3 > 0.f ? r3 : r3 * res_inv;
                    }
                    v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
                }
                if (p.out_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ups_vmax(v[e], oact_ns * v[e]);
                }
                *(uint2*)(stage + p16 * 80 + (16 * j + 4 * q16) * 2) = make_uint2(Chunk<T>::pk(v[0], v[1]), Chunk<T>::pk(v[2], v[3]));
            }
            const uint4 o = *(const uint4*)(stage + (lane >> 2) * 80 + (lane & 3) * 16);
            *(uint4*)(out_img + ((long long)y * W + 16 * ct + (lane >> 2)) * p.ldo * 2 + (cg * 32 + 8 * (lane & 3)) * 2) = o;
            // (no sign bytes from this kernel: its outputs feed `downsample` convolutions, which apply no activation to their input.
            // -DUPS_ROWS_FWD_SIGN adds the conditional store that made the 64-channel instance irreproducible in round 5; together with
            // -DUPS_ROWS_NO_FENCE and packed fp32 instructions allowed it is the REPRODUCER of docs/design/rows_hazard.md, and with this
            // file's two rules in force it is the regression build of tools/probes/rows_hunt.sh: 0 wrong launches)
#ifdef UPS_ROWS_FWD_SIGN
            if (p.sign_out)
                p.sign_out[(((long long)img * p.h + y) * W + 16 * ct + (lane >> 2)) * (p.ldo >> 3) + cg * 4 + (lane & 3)] = (unsigned char)ups_sign_byte(o);
#endif
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (outstanding row requests past the band target this block's LDS)
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same stream with TWO column tiles per wave: 16 (column tile, channel group) jobs per row instead of 8 -- 64 channels at 128
// columns (VGG block 1, the hourglass decoder, encoder_1's second residual block of the 256x256 configs) and 32 channels at 256 columns
// (its first).  Two DMA pieces per wave and row, one block per CU (133 KB of row ring).  Forward-type epilogue (bias, residual from the
// centre row, stored activation); FLIP for input gradients without an activation derivative.  (The input gradient WITH act' of these
// shapes takes the patch kernel, which reads the producer's sign bytes: the act' form of this kernel brought its operand in by
// inline-asm register loads that hipcc moved ahead of their counted wait -- off since round 5, deleted in round 6.)
template <typename T, int CI, int LW, int NR, bool FLIP>
__global__ __launch_bounds__(512, 2) void conv3x3_rows2_kernel(const RowsK p) {
    constexpr int W = 1 << LW, KC = CI / 32, NCT = W / 16, NCP = NCT / 2;       // column tiles, column-tile pairs
    constexpr int PL = (W + 2) * 64, RB = KC * PL, ST = 16 * 80;
    constexpr int L = (NR - 4) / 2;
    constexpr int GS = 4;                       // DMA instructions (= stores) a wave issues per iteration
    static_assert(KC * NCT == 16 && (8 / NCP) * NCP == 8, "two DMA pieces per wave and row");
    typedef typename RFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem + NR * RB + wid * ST;
    float* biasL = (float*)(smem + NR * RB + 8 * ST);
    const int p16 = lane & 15, q16 = lane >> 4;
    const int ctp = wid % NCP, cg = wid / NCP;               // column-tile pair, output-channel group
    const int band = blockIdx.x % p.bands, img = blockIdx.x / p.bands;
    const int y0 = band * p.band_rows;
    const int y1 = min(p.h, y0 + p.band_rows);
    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);
    frag_t wb[9][KC][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[t][kc][j] = *(const frag_t*)(p.w + ((long long)(t * KC + kc) * p.co_tot + cg * 32 + 16 * j + p16) * 64 + q16 * 16);
    if (tid < 64) biasL[tid] = (p.bias && tid < p.co_tot) ? p.bias[tid] : 0.f;
    __syncthreads();

    // ---- row DMA: pieces 2 wid, 2 wid + 1 of the KC * NCT pieces of a row: plane = piece / NCT, 16-pixel segment = piece % NCT
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int pl0 = (2 * wid) / NCT, sg0 = (2 * wid) % NCT, pl1 = (2 * wid + 1) / NCT, sg1 = (2 * wid + 1) % NCT;
    const int Pd0 = 1 + 16 * sg0 + (lane >> 2), Pd1 = 1 + 16 * sg1 + (lane >> 2);
    const unsigned d_off0 = (unsigned)((16 * sg0 + (lane >> 2)) * p.ldi * 2 + pl0 * 64 + (((lane & 3) ^ r_swz(Pd0)) << 4));
    const unsigned d_off1 = (unsigned)((16 * sg1 + (lane >> 2)) * p.ldi * 2 + pl1 * 64 + (((lane & 3) ^ r_swz(Pd1)) << 4));
    const unsigned char* in_img = p.in + (long long)img * p.h * W * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(W * p.ldi * 2);
    auto issue_row = [&](int k) __attribute__((always_inline)) {
        const int y = y0 - 1 + k;
        const unsigned char* src = (unsigned)y < (unsigned)p.h ? in_img + (long long)y * row_bytes : ups_rows_zero;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + pl0 * PL + (1 + 16 * sg0) * 64));
        const unsigned dst1 = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + pl1 * PL + (1 + 16 * sg1) * 64));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst0), "v"(d_off0), "s"(src) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst1), "v"(d_off1), "s"(src) : "memory", "m0");
    };
    issue_row(0); issue_row(1);
#pragma unroll
    for (int i = 0; i < L; ++i) { issue_row(2 * i + 2); issue_row(2 * i + 3); }

    const float oact_ns = ups_slope_eff(p.out_act, p.slope);
    const float res_inv = p.res_act ? 1.f / p.slope : 1.f;
    const int iters = (y1 - y0 + 1) >> 1;
    unsigned char* out_img = p.out + (long long)img * p.h * W * p.ldo * 2;
    for (int it = 0; it < iters; ++it) {
        // (as conv3x3_rows_kernel; per wave and iteration four row requests and four stores.  The requests of iteration it - L must
        // have landed: younger are the other prologue requests (it == 0) or the later iterations' requests and this count of stores)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * GS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * GS + 4) : "memory");
        __builtin_amdgcn_s_barrier();
        const int yb = y0 + 2 * it;
        issue_row(2 * it + 2 * L + 2);
        issue_row(2 * it + 2 * L + 3);
        const unsigned char* rowp[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) rowp[r4] = ring + ((2 * it + r4) % NR) * RB;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ct = 2 * ctp + c;
            f32x4v acc[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4v b4 = *(const f32x4v*)(biasL + cg * 32 + 16 * j + 4 * q16);
                acc[0][j] = b4; acc[1][j] = b4;
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int P = 16 * ct + p16 + dx;
                const int aoff = P * 64 + ((q16 ^ r_swz(P)) << 4);
                frag_t a[4][KC];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) a[r4][kc] = *(const frag_t*)(rowp[r4] + kc * PL + aoff);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
                        const int t = FLIP ? (2 - dyi) * 3 + (2 - dx) : dyi * 3 + dx;
#pragma unroll
                        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                if constexpr (__is_same(T, bf16))
                                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[t][kc][j], a[r + dyi][kc], acc[r][j], 0, 0, 0);
                                else
                                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[t][kc][j], a[r + dyi][kc], acc[r][j], 0, 0, 0);
                            }
                    }
            }
            ups_rows_fence();
            const int Pc = 16 * ct + p16 + 1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = min(yb + r, p.h - 1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[r][j][e];
                    if (p.res_self) {
                        const int c32 = 16 * j + 4 * q16;
                        const uint2 rr = *(const uint2*)(rowp[1 + r] + cg * PL + Pc * 64 + (((c32 >> 3) ^ r_swz(Pc)) << 4) + (q16 & 1) * 8);
                        float r0, r1, r2, r3;
                        ups_unpack2<T>(rr.x, r0, r1); ups_unpack2<T>(rr.y, r2, r3);
                        if (p.res_act) {
                            r0 = r0 > 0.f ? r0 : r0 * res_inv; r1 = r1 > 0.f ? r1 : r1 * res_inv;
                            r2 = r2 > 0.f ? r2 : r2 * res_inv; r3 = r3 > 0.f ? r3 : r3 * res_inv;
                        }
                        v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3;
                    }
                    if (p.out_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ups_vmax(v[e], oact_ns * v[e]);
                    }
                    *(uint2*)(stage + p16 * 80 + (16 * j + 4 * q16) * 2) = make_uint2(Chunk<T>::pk(v[0], v[1]), Chunk<T>::pk(v[2], v[3]));
                }
                // (band heights are even: every wave stores twice per column tile -- a static count for the waits above)
                const uint4 o = *(const uint4*)(stage + (lane >> 2) * 80 + (lane & 3) * 16);
                *(uint4*)(out_img + ((long long)y * W + 16 * ct + (lane >> 2)) * p.ldo * 2 + (cg * 32 + 8 * (lane & 3)) * 2) = o;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, int CI, int LW, int NR, bool FLIP>
int launch_rows2(const RowsK& k, hipStream_t s) {
    constexpr int W = 1 << LW, KC = CI / 32;
    constexpr size_t smem = (size_t)NR * KC * (W + 2) * 64 + 8 * 16 * 80 + 256;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_rows2_kernel<T, CI, LW, NR, FLIP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows2_kernel<T, CI, LW, NR, FLIP>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    return UPS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Input gradient of the part-masked FIRST convolution of encoder_1 (cub/code/SB_model48i/model.py:176-187: the P x B part images
// are view[b] * hard[b, :, :, p], never materialised), reduced to the hard mask on the way out:
//     g_hard[b][y][x][p] = sum_c bf16(gx[p * B + b][y][x][c]) * view[b][y][x][c],   gx = conv^T(gy)  (3 view channels),
// the contraction conv3x3_patch.hip's mask_grad epilogue does per 16x16 tile (0.39 ms for a 0.67 GB read).  Row stream of the
// 32-channel gradient tensor, one 16-channel output block per wave (channels 0 .. 2 used), flipped taps; lanes q16 == 0 hold the
// pixel's three channels and write one float each.
struct MaskGK {
    const unsigned char* in; const unsigned char* w; const float* view; float* g_hard;
    int n, h, ldi, co, B, P, band_rows, bands;
};

// CTW column tiles per wave: 1 at 128 columns (two blocks per CU), 2 at 256 (one).
template <int NR, int LW, int CTW>
__global__ __launch_bounds__(512, CTW == 1 ? 4 : 2) void conv3x3_rows_maskgrad_kernel(const MaskGK p) {
    constexpr int W = 1 << LW, PL = (W + 2) * 64, RB = PL;
    constexpr int L = (NR - 4) / 2;
    static_assert(L == 2 && W == 128 * CTW, "the counted waits below are written for a lead of two iterations; 16 CTW columns per wave");
    constexpr int NOP = 6 * CTW;               // vector-memory operations per wave and iteration: 2 CTW view loads, 2 CTW requests, 2 CTW stores
    typedef bf16x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);         // column tile (pair) = DMA segment (pair)
    const int p16 = lane & 15, q16 = lane >> 4;
    const int band = blockIdx.x % p.bands, img = blockIdx.x / p.bands;       // img = part * B + b (part-major)
    const int part = img / p.B, b = img - part * p.B;
    const int y0 = band * p.band_rows, y1 = min(p.h, y0 + p.band_rows);
    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);
    frag_t wb[9];
    const int wrow = min(p16, p.co - 1);         // rows past co repeat the last one; those outputs are not used
#pragma unroll
    for (int t = 0; t < 9; ++t) wb[t] = *(const frag_t*)(p.w + ((long long)t * p.co + wrow) * 64 + q16 * 16);
    __syncthreads();
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned d_off[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int ct = CTW * wid + c;
        const int Pd = 1 + 16 * ct + (lane >> 2);
        d_off[c] = (unsigned)((16 * ct + (lane >> 2)) * p.ldi * 2 + (((lane & 3) ^ r_swz(Pd)) << 4));
    }
    const unsigned char* in_img = p.in + (long long)img * p.h * W * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(W * p.ldi * 2);
    auto issue_row = [&](int k) __attribute__((always_inline)) {
        const int y = y0 - 1 + k;
        const unsigned char* src = (unsigned)y < (unsigned)p.h ? in_img + (long long)y * row_bytes : ups_rows_zero;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + (1 + 16 * (CTW * wid + c)) * 64));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(d_off[c]), "s"(src) : "memory", "m0");
        }
    };
    issue_row(0); issue_row(1);
#pragma unroll
    for (int i = 0; i < L; ++i) { issue_row(2 * i + 2); issue_row(2 * i + 3); }
    const float* view_img = p.view + (long long)b * p.h * W * p.co;
    float* gh_img = p.g_hard + (long long)b * p.h * W * p.P + part;
    const int iters = (y1 - y0 + 1) >> 1;
    for (int it = 0; it < iters; ++it) {
        // Per iteration a wave issues, in this order: 2 CTW view loads (the three fp32 view channels of its pixels in the two output
        // rows; inline asm like the DMAs -- a compiler-visible load would be waited for with vmcnt(0), i.e. together with every row
        // request in flight), 2 CTW row requests, 2 CTW stores.  The requests of iteration it - L must have landed here: younger
        // operations are the other prologue requests (it == 0) or at least the previous iteration's NOP (it >= 1).
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * 2 * CTW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * NOP) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x3v vw[CTW][2];
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = min(y0 + 2 * it + r, p.h - 1);
                const float* vp = view_img + ((long long)y * W + 16 * (CTW * wid + c) + p16) * p.co;
                asm volatile("global_load_dwordx3 %0, %1, off" : "=&v"(vw[c][r]) : "v"(vp) : "memory");
            }
        issue_row(2 * it + 2 * L + 2);
        issue_row(2 * it + 2 * L + 3);
        const unsigned char* rowp[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) rowp[r4] = ring + ((2 * it + r4) % NR) * RB;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int ct = CTW * wid + c;
            f32x4v acc[2];
            acc[0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int P = 16 * ct + p16 + dx;
                const int aoff = P * 64 + ((q16 ^ r_swz(P)) << 4);
                frag_t a[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) a[r4] = *(const frag_t*)(rowp[r4] + aoff);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[(2 - dyi) * 3 + (2 - dx)], a[r + dyi], acc[r], 0, 0, 0);
            }
            ups_rows_fence();
            // the view values: behind them the 2 CTW row requests (and the first tile's two stores) may stay in flight
            if (c == 0) {
                if constexpr (CTW == 1) asm volatile("s_waitcnt vmcnt(2)" : "+v"(vw[0][0]), "+v"(vw[0][1]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" : "+v"(vw[0][0]), "+v"(vw[0][1]), "+v"(vw[CTW - 1][0]), "+v"(vw[CTW - 1][1]) :: "memory");
            }
            const int x = 16 * ct + p16;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = y0 + 2 * it + r;
                float s = (float)(bf16)acc[r][0] * vw[c][r][0];
                s += (float)(bf16)acc[r][1] * vw[c][r][1];
                s += (float)(bf16)acc[r][2] * vw[c][r][2];
                // (every wave stores in every iteration -- a static count for the waits above; rows past the band and the lanes that
                // hold other channels write nothing)
                float* dst = gh_img + ((long long)min(y, p.h - 1) * W + x) * p.P;
                const bool on = y < y1 && q16 == 0;
                asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, -1" :: "s"(__ballot(on)), "v"(dst), "v"(s) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same stream for the 3x3 / STRIDE-2 `downsample` forwards (cub/code/nn.py:816-817) of even-sized images ('SAME': taps at input
// rows 2 y .. 2 y + 2, columns 2 x .. 2 x + 2, zero beyond the image).  A row buffer holds the even and the odd pixels of an input row
// in separate planes [32-channel plane][parity][slot = x / 2][64 B] (+ one zero slot behind the even plane: pixel W of the last column
// tap), so the three column taps of 16 consecutive OUTPUT pixels are unit-stride fragment reads (even plane at slot x, odd plane at
// slot x, even plane at slot x + 1) with the same swizzle as above; a DMA piece fetches 16 same-parity pixels (64-byte pieces at a
// 128-byte stride: the two parities of a row together read every line once).  One output row per iteration (three input rows, two of
// them new); 8 waves = (W_out / 16 column tiles) x (groups of 32 outputs): 64 outputs at W_in = 128, 128 at W_in = 64 -- the encoders'
// first two downsample layers exactly.
template <int CI, int LWI, int NR>
__global__ __launch_bounds__(512, CI == 32 ? 4 : 2) void conv3x3_rows_s2_kernel(const RowsK p) {
    typedef bf16 T;
    constexpr int WI = 1 << LWI, WO = WI / 2, KC = CI / 32, NCT = WO / 16, NCG = 8 / NCT, SG = WI / 32;
    constexpr int PLS = (WI / 2 + 1) * 64;      // bytes of one parity plane
    constexpr int RB = KC * 2 * PLS;
    constexpr int ST = 16 * 80;
    constexpr int L = (NR - 3) / 2;             // NR = 2 L + 3: rows 2 it .. 2 it + 2 in use, 2 L requested ahead
    static_assert(KC * 2 * SG == 8, "one DMA piece per wave and input row");
    typedef bf16x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem + NR * RB + wid * ST;
    float* biasL = (float*)(smem + NR * RB + 8 * ST);         // 128 floats
    const int p16 = lane & 15, q16 = lane >> 4;
    const int ct = wid % NCT, cg = wid / NCT;
    const int seg = wid % SG, par = (wid / SG) & 1, kcp = wid / (2 * SG);
    const int band = blockIdx.x % p.bands, img = blockIdx.x / p.bands;
    const int ho = p.h / 2;
    const int y0 = band * p.band_rows;                        // output rows [y0, y1)
    const int y1 = min(ho, y0 + p.band_rows);

    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);
    frag_t wb[9][KC][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[t][kc][j] = *(const frag_t*)(p.w + ((long long)(t * KC + kc) * p.co_tot + cg * 32 + 16 * j + p16) * 64 + q16 * 16);
    if (tid < 128) biasL[tid] = (p.bias && tid < p.co_tot) ? p.bias[tid] : 0.f;
    __syncthreads();

    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int Pd = 16 * seg + (lane >> 2);                    // slot of the lane's pixel 2 Pd + par
    const unsigned d_off = (unsigned)((2 * Pd + par) * p.ldi * 2 + kcp * 64 + (((lane & 3) ^ r_swz(Pd)) << 4));
    const unsigned char* in_img = p.in + (long long)img * p.h * WI * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(WI * p.ldi * 2);
    auto issue_row = [&](int k) __attribute__((always_inline)) {          // input row 2 y0 + k
        const int y = 2 * y0 + k;
        const unsigned char* src = y < p.h ? in_img + (long long)y * row_bytes : ups_rows_zero;
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + (kcp * 2 + par) * PLS + 16 * seg * 64));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(d_off), "s"(src) : "memory", "m0");
    };
    issue_row(0);
#pragma unroll
    for (int i = 0; i < L; ++i) { issue_row(2 * i + 1); issue_row(2 * i + 2); }

    const float oact_ns = ups_slope_eff(p.out_act, p.slope);
    unsigned char* out_img = p.out + (long long)img * ho * WO * p.ldo * 2;
    const int iters = y1 - y0;
    for (int it = 0; it < iters; ++it) {
        // the requests of iteration it - L (input rows up to 2 it + 2) must have landed; younger: two requests per later iteration
        // and, from the first real iteration on, one store each (counted once)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * 2 + 1) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_row(2 * (it + L) + 1);
        issue_row(2 * (it + L) + 2);

        f32x4v acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = *(const f32x4v*)(biasL + cg * 32 + 16 * j + 4 * q16);
#pragma unroll
        for (int dyi = 0; dyi < 3; ++dyi) {
            const unsigned char* rowp = ring + ((2 * it + dyi) % NR) * RB;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int P = 16 * ct + p16 + (dx == 2 ? 1 : 0);
                const int aoff = (dx & 1) * PLS + P * 64 + ((q16 ^ r_swz(P)) << 4);
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const frag_t a = *(const frag_t*)(rowp + kc * 2 * PLS + aoff);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[dyi * 3 + dx][kc][j], a, acc[j], 0, 0, 0);
                }
            }
        }
        ups_rows_fence();
        const int y = y0 + it;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[j][e];
            if (p.out_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ups_vmax(v[e], oact_ns * v[e]);
            }
            *(uint2*)(stage + p16 * 80 + (16 * j + 4 * q16) * 2) = make_uint2(Chunk<T>::pk(v[0], v[1]), Chunk<T>::pk(v[2], v[3]));
        }
        const uint4 o = *(const uint4*)(stage + (lane >> 2) * 80 + (lane & 3) * 16);
        *(uint4*)(out_img + ((long long)y * WO + 16 * ct + (lane >> 2)) * p.ldo * 2 + (cg * 32 + 8 * (lane & 3)) * 2) = o;
        if (p.sign_out)
            p.sign_out[(((long long)img * ho + y) * WO + 16 * ct + (lane >> 2)) * (p.ldo >> 3) + cg * 4 + (lane & 3)] = (unsigned char)ups_sign_byte(o);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Two column tiles per wave (16 jobs per output row): 32 -> 64 channels at 256 input columns, 64 -> 128 at 128 -- the same two layers
// of the 256 x 256 configs.  Two DMA pieces per wave and input row, one block per CU.
template <int CI, int LWI, int NR>
__global__ __launch_bounds__(512, 2) void conv3x3_rows_s2x2_kernel(const RowsK p) {
    typedef bf16 T;
    constexpr int WI = 1 << LWI, WO = WI / 2, KC = CI / 32, NCT = WO / 16, NCP = NCT / 2, SG = WI / 32;
    constexpr int PLS = (WI / 2 + 1) * 64;
    constexpr int RB = KC * 2 * PLS;
    constexpr int ST = 16 * 80;
    constexpr int L = (NR - 3) / 2;
    static_assert(KC * 2 * SG == 16, "two DMA pieces per wave and input row");
    typedef bf16x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* stage = smem + NR * RB + wid * ST;
    float* biasL = (float*)(smem + NR * RB + 8 * ST);         // 128 floats
    const int p16 = lane & 15, q16 = lane >> 4;
    const int ctp = wid % NCP, cg = wid / NCP;
    const int band = blockIdx.x % p.bands, img = blockIdx.x / p.bands;
    const int ho = p.h / 2;
    const int y0 = band * p.band_rows;
    const int y1 = min(ho, y0 + p.band_rows);
    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);
    frag_t wb[9][KC][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                wb[t][kc][j] = *(const frag_t*)(p.w + ((long long)(t * KC + kc) * p.co_tot + cg * 32 + 16 * j + p16) * 64 + q16 * 16);
    if (tid < 128) biasL[tid] = (p.bias && tid < p.co_tot) ? p.bias[tid] : 0.f;
    __syncthreads();

    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const int pi0 = 2 * wid, pi1 = 2 * wid + 1;
    const int sg0 = pi0 % SG, pr0 = (pi0 / SG) & 1, kp0 = pi0 / (2 * SG), sg1 = pi1 % SG, pr1 = (pi1 / SG) & 1, kp1 = pi1 / (2 * SG);
    const int Pd0 = 16 * sg0 + (lane >> 2), Pd1 = 16 * sg1 + (lane >> 2);
    const unsigned d_off0 = (unsigned)((2 * Pd0 + pr0) * p.ldi * 2 + kp0 * 64 + (((lane & 3) ^ r_swz(Pd0)) << 4));
    const unsigned d_off1 = (unsigned)((2 * Pd1 + pr1) * p.ldi * 2 + kp1 * 64 + (((lane & 3) ^ r_swz(Pd1)) << 4));
    const unsigned char* in_img = p.in + (long long)img * p.h * WI * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(WI * p.ldi * 2);
    auto issue_row = [&](int k) __attribute__((always_inline)) {          // input row 2 y0 + k
        const int y = 2 * y0 + k;
        const unsigned char* src = y < p.h ? in_img + (long long)y * row_bytes : ups_rows_zero;
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + (kp0 * 2 + pr0) * PLS + 16 * sg0 * 64));
        const unsigned dst1 = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((k % NR) * RB + (kp1 * 2 + pr1) * PLS + 16 * sg1 * 64));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst0), "v"(d_off0), "s"(src) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst1), "v"(d_off1), "s"(src) : "memory", "m0");
    };
    issue_row(0);
#pragma unroll
    for (int i = 0; i < L; ++i) { issue_row(2 * i + 1); issue_row(2 * i + 2); }

    const float oact_ns = ups_slope_eff(p.out_act, p.slope);
    unsigned char* out_img = p.out + (long long)img * ho * WO * p.ldo * 2;
    const int iters = y1 - y0;
    for (int it = 0; it < iters; ++it) {
        // (four requests and two stores per wave and iteration)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * 4 + 2) : "memory");
        __builtin_amdgcn_s_barrier();
        issue_row(2 * (it + L) + 1);
        issue_row(2 * (it + L) + 2);
        const int y = y0 + it;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ct = 2 * ctp + c;
            f32x4v acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = *(const f32x4v*)(biasL + cg * 32 + 16 * j + 4 * q16);
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const unsigned char* rowp = ring + ((2 * it + dyi) % NR) * RB;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int P = 16 * ct + p16 + (dx == 2 ? 1 : 0);
                    const int aoff = (dx & 1) * PLS + P * 64 + ((q16 ^ r_swz(P)) << 4);
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) {
                        const frag_t a = *(const frag_t*)(rowp + kc * 2 * PLS + aoff);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[dyi * 3 + dx][kc][j], a, acc[j], 0, 0, 0);
                    }
                }
            }
            ups_rows_fence();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][e];
                if (p.out_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ups_vmax(v[e], oact_ns * v[e]);
                }
                *(uint2*)(stage + p16 * 80 + (16 * j + 4 * q16) * 2) = make_uint2(Chunk<T>::pk(v[0], v[1]), Chunk<T>::pk(v[2], v[3]));
            }
            const uint4 o = *(const uint4*)(stage + (lane >> 2) * 80 + (lane & 3) * 16);
            *(uint4*)(out_img + ((long long)y * WO + 16 * ct + (lane >> 2)) * p.ldo * 2 + (cg * 32 + 8 * (lane & 3)) * 2) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CI, int LWI, int NR>
int launch_rows_s2x2(const RowsK& k, hipStream_t s) {
    constexpr int WI = 1 << LWI, KC = CI / 32;
    constexpr size_t smem = (size_t)NR * KC * 2 * (WI / 2 + 1) * 64 + 8 * 16 * 80 + 512;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_rows_s2x2_kernel<CI, LWI, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows_s2x2_kernel<CI, LWI, NR>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    return UPS_OK;
}

template <int CI, int LWI, int NR>
int launch_rows_s2(const RowsK& k, hipStream_t s) {
    constexpr int WI = 1 << LWI, KC = CI / 32;
    constexpr size_t smem = (size_t)NR * KC * 2 * (WI / 2 + 1) * 64 + 8 * 16 * 80 + 512;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_rows_s2_kernel<CI, LWI, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows_s2_kernel<CI, LWI, NR>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    return UPS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K-deep, thin-out form: the mask decoder's logit convolution 256 (+2 CoordConv) -> P <= 16 channels (cub/code/SB_model48i/model.py:154),
// a 1.07 GB read for 84 MB of output.  The patch kernel ran it with one 32-channel patch chunk in flight per block -- 20 us of
// dependent HBM round trips per tile (tools/probes/phase_timing.py), 2 TB/s.  Here a block owns a strip of 32 columns x 32 rows and
// its eight waves split K: wave w streams the rows of ITS 32-channel plane (34 pixel slots with the halo) through a private ring --
// its own DMA pieces, its own counted waits, no barrier for the input at all -- with the nine taps' weights of that plane in
// registers (36 VGPRs), and computes a partial 32 pixels x 16 channels per output row (18 MFMAs for 18 fragment reads).  The eight
// partials meet in LDS (double-buffered: one barrier per output row), thread (pixel, channel) adds them in a fixed order together with
// bias and the CoordConv class-table term and stores fp32 or 16-bit.
struct ThinK {
    const unsigned char* in; const unsigned char* w; unsigned char* out;
    const float* bias; const float* coord_tab;
    int n, h, wd, ldi, ldo, co, co_fill, out_f32, strips, bands, band_rows;
};

// (no lambdas and no local arrays besides fragments in this kernel: with closures capturing by reference hipcc kept the argument struct
// and the epilogue constants in scratch, and every scratch access is a vector-memory operation whose wait drains the row requests)
#define UPS_THIN_DMA(SRC, DST, OFF, MLO, MHI)                                                                                      \
    do {                                                                                                                         \
        const unsigned long long msk_ = ((unsigned long long)(MHI) << 32) | (MLO);                                               \
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"      \
                     :: "s"(msk_), "s"(DST), "v"(OFF), "s"(SRC) : "memory", "m0");                                               \
    } while (0)
#define UPS_THIN_ISSUE_ROW(K)                                                                                                    \
    do {                                                                                                                         \
        const int yy_ = y0 - 1 + (K);                                                                                            \
        const unsigned char* s0_ = (unsigned)yy_ < (unsigned)h_ ? in_img + (long long)yy_ * row_bytes : ups_rows_zero;           \
        const unsigned long long sa_ = (unsigned long long)s0_;                                                                  \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa_ >> 32));      /* (readfirstlane returns int) */   \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa_);                                            \
        const unsigned char* src_ = (const unsigned char*)(((unsigned long long)hi_ << 32) | (unsigned long long)lo_);          \
        const unsigned base_ = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(((K) % NR) * RB + wid * PLW));                \
        UPS_THIN_DMA(src_, base_, d_off0, m0lo, m0hi);                                                                           \
        UPS_THIN_DMA(src_, base_ + 1024u, d_off1, m1lo, m1hi);                                                                   \
        if constexpr (NPC > 2) UPS_THIN_DMA(src_, base_ + 2048u, d_off2, m2lo, m2hi);                                            \
    } while (0)

template <typename T, int NR, int SW>        // SW: strip width (32: one block per CU; 16: two)
__global__ __launch_bounds__(512, SW == 16 ? 4 : 2) void conv3x3_thinout_kernel(const ThinK p) {
    constexpr int NSL = SW + 2, NC2 = SW / 16, NPC = (NSL + 15) / 16;        // pixel slots per plane row, column tiles, DMA pieces
    constexpr int PLW = NSL * 64;               // bytes of one plane of a row
    constexpr int RB = 8 * PLW;                 // eight planes = 256 channels
    constexpr int L = NR - 3;                   // rows it .. it + 2 in use, L requested ahead
    constexpr int RED = SW * 20;                // floats of one wave's partial row (pixel pitch 20 floats)
    typedef typename RFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* red = (float*)(smem + NR * RB);      // [2][8][RED]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p16 = lane & 15, q16 = lane >> 4;
    // the argument struct's fields used inside the loop, as plain locals
    const int h_ = p.h, wd_ = p.wd, ldo_ = p.ldo, co_ = p.co, cofill_ = p.co_fill, of32_ = p.out_f32;
    const bool has_ct = p.coord_tab != nullptr;
    int t = blockIdx.x;
    const int strip = t % p.strips; t /= p.strips;
    const int band = t % p.bands; const int img = t / p.bands;
    const int x0 = strip * SW, y0 = band * p.band_rows;
    const int y1 = min(h_, y0 + p.band_rows);

    for (int i = tid * 16; i < NR * RB; i += 512 * 16) *(uint4*)(ring + i) = make_uint4(0u, 0u, 0u, 0u);
    // the wave's weights: plane wid, co rows 0 .. 15 (rows past co repeat the last one; their outputs are never stored)
    frag_t wb0, wb1, wb2, wb3, wb4, wb5, wb6, wb7, wb8;
    {
        const int wrow = min(p16, co_ - 1);
        const unsigned char* wp = p.w + ((long long)wid * co_ + wrow) * 64 + q16 * 16;
        const long long ts = (long long)8 * co_ * 64;
        wb0 = *(const frag_t*)(wp); wb1 = *(const frag_t*)(wp + ts); wb2 = *(const frag_t*)(wp + 2 * ts);
        wb3 = *(const frag_t*)(wp + 3 * ts); wb4 = *(const frag_t*)(wp + 4 * ts); wb5 = *(const frag_t*)(wp + 5 * ts);
        wb6 = *(const frag_t*)(wp + 6 * ts); wb7 = *(const frag_t*)(wp + 7 * ts); wb8 = *(const frag_t*)(wp + 8 * ts);
    }
    // the thread's output element in the reduction: (pixel px_t, channel co_t) of the current row; its bias and CoordConv terms for the
    // three row classes (first row / interior / last row) of its column class
    const int co_t = tid & 15, px_t = (tid >> 4) & (SW - 1);
    const bool t_on = tid < SW * 16;
    const int xg = x0 + px_t;
    const bool c_ok = co_t < co_;
    float bias_l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (c_ok) {
        if (p.bias) bias_l = p.bias[co_t];
        if (has_ct) {
            const int xm = (xg > 0 ? 1 : 0) | 2 | (xg + 1 < wd_ ? 4 : 0);
            const float* ta = p.coord_tab + (long long)((2 | 4) * 8 + xm) * 3 * co_ + co_t;        // first row of the image
            const float* tb = p.coord_tab + (long long)(7 * 8 + xm) * 3 * co_ + co_t;              // interior rows
            const float* tc = p.coord_tab + (long long)((1 | 2) * 8 + xm) * 3 * co_ + co_t;        // last row
            a0 = ta[0]; a1 = ta[co_]; a2 = ta[2 * co_];
            b0 = tb[0]; b1 = tb[co_]; b2 = tb[2 * co_];
            c0 = tc[0]; c1 = tc[co_]; c2 = tc[2 * co_];
        }
    }
    // every set-up load is waited for HERE, before the first row request: left to the compiler the waits would sit at the first uses
    // inside the loop (vmcnt(0) in every iteration)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias_l), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2)
                 :: "memory");
    const float bias_t = bias_l, ct0a = a0, ct1a = a1, ct2a = a2, ct0b = b0, ct1b = b1, ct2b = b2, ct0c = c0, ct1c = c1, ct2c = c2;
    __syncthreads();

    // ---- wave-private row DMA: pieces of 16 pixel slots of plane wid
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned d_off0 = 0u, d_off1 = 0u, d_off2 = 0u, m0lo = 0u, m0hi = 0u, m1lo = 0u, m1hi = 0u, m2lo = 0u, m2hi = 0u;
#define UPS_THIN_PIECE(Q, OFF, MLO, MHI)                                                                                          \
    do {                                                                                                                         \
        const int P_ = 16 * (Q) + (lane >> 2), x_ = x0 - 1 + P_;                                                                 \
        const bool ok_ = P_ < NSL && (unsigned)x_ < (unsigned)wd_;                                                               \
        OFF = ok_ ? (unsigned)(x_ * p.ldi * 2 + wid * 64 + (((lane & 3) ^ r_swz(P_)) << 4)) : 0u;                                \
        const unsigned long long m_ = __ballot(ok_);                                                                            \
        MLO = __builtin_amdgcn_readfirstlane((unsigned)m_); MHI = __builtin_amdgcn_readfirstlane((unsigned)(m_ >> 32));          \
    } while (0)
    UPS_THIN_PIECE(0, d_off0, m0lo, m0hi);
    UPS_THIN_PIECE(1, d_off1, m1lo, m1hi);
    if constexpr (NPC > 2) UPS_THIN_PIECE(2, d_off2, m2lo, m2hi);
#undef UPS_THIN_PIECE
    const unsigned char* in_img = p.in + (long long)img * h_ * wd_ * p.ldi * 2;
    const unsigned row_bytes = (unsigned)(wd_ * p.ldi * 2);
#pragma unroll
    for (int k = 0; k < L + 2; ++k) UPS_THIN_ISSUE_ROW(k);

    const int iters = y1 - y0;
    unsigned char* out_img = p.out + (long long)img * h_ * wd_ * ldo_ * (of32_ ? 4 : 2);
    // (round 6, late) the fragments of a plane row are read from LDS ONCE, when the row enters as row it + 2, and stay in registers for the
    // three output rows that use it (slot = row % 3: the loop is unrolled by three so that the slots are static): 6 fragment reads per
    // output row instead of 18 -- the ring's LDS reads were 147 KB per output row and CU
    frag_t fr[3][3][NC2];
#define UPS_THIN_LOAD_ROW(ROWI, SL)                                                                                              \
    do {                                                                                                                         \
        const unsigned char* rowp_ = ring + ((ROWI) % NR) * RB + wid * PLW;                                                      \
        _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_)                                                                      \
            _Pragma("unroll") for (int c2i = 0; c2i < NC2; ++c2i) {                                                              \
                const int P_ = 16 * c2i + p16 + dx_;                                                                             \
                fr[SL][dx_][c2i] = *(const frag_t*)(rowp_ + P_ * 64 + ((q16 ^ r_swz(P_)) << 4));                                 \
            }                                                                                                                    \
    } while (0)
#define UPS_THIN_MMA(WB, SL, DX)                                                                                                 \
    do {                                                                                                                         \
        _Pragma("unroll") for (int c2i = 0; c2i < NC2; ++c2i) {                                                                  \
            if constexpr (__is_same(T, bf16)) acc[c2i][tpar] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WB, fr[SL][DX][c2i], acc[c2i][tpar], 0, 0, 0); \
            else acc[c2i][tpar] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WB, fr[SL][DX][c2i], acc[c2i][tpar], 0, 0, 0);         \
        }                                                                                                                        \
        tpar ^= 1;                                                                                                               \
    } while (0)
#define UPS_THIN_BODY(IT, S0, S1, S2)                                                                                            \
    do {                                                                                                                         \
        const int it = (IT);                                                                                                     \
        /* rows it .. it + 2 of this wave's plane must have landed (row it + 2 was requested by iteration it - L); younger: the NPC */ \
        /* requests of each of the L - 1 rows behind it and the stores issued since (every iteration ends with one: counted once) */ \
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * NPC) : "memory");                                       \
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((L - 1) * NPC + 1) : "memory");                                           \
        UPS_THIN_ISSUE_ROW(it + L + 2);      /* into the slot of row it - 1 (in registers since iteration it - 3) */             \
        if (it == 0) { UPS_THIN_LOAD_ROW(0, S0); UPS_THIN_LOAD_ROW(1, S1); }                                                     \
        UPS_THIN_LOAD_ROW(it + 2, S2);                                                                                           \
        f32x4v acc[NC2][2];                                                                                                      \
        _Pragma("unroll") for (int c2i = 0; c2i < NC2; ++c2i) { acc[c2i][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[c2i][1] = acc[c2i][0]; } \
        int tpar = 0;                                                                                                            \
        UPS_THIN_MMA(wb0, S0, 0); UPS_THIN_MMA(wb1, S0, 1); UPS_THIN_MMA(wb2, S0, 2);                                            \
        UPS_THIN_MMA(wb3, S1, 0); UPS_THIN_MMA(wb4, S1, 1); UPS_THIN_MMA(wb5, S1, 2);                                            \
        UPS_THIN_MMA(wb6, S2, 0); UPS_THIN_MMA(wb7, S2, 1); UPS_THIN_MMA(wb8, S2, 2);                                            \
        float* mine = red + ((it & 1) * 8 + wid) * RED;                                                                          \
        _Pragma("unroll") for (int c2i = 0; c2i < NC2; ++c2i) *(f32x4v*)(mine + (16 * c2i + p16) * 20 + 4 * q16) = acc[c2i][0] + acc[c2i][1]; \
        /* (a raw barrier: __syncthreads() carries a fence that drains vmcnt -- every row request in flight -- each iteration) */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                            \
        /* thread (px_t, co_t): the eight partials of this output row, in wave order */                                          \
        const float* all = red + (it & 1) * 8 * RED + px_t * 20 + co_t;                                                          \
        float v = all[0];                                                                                                        \
        _Pragma("unroll") for (int w8 = 1; w8 < 8; ++w8) v += all[w8 * RED];                                                     \
        const int y = y0 + it;                                                                                                   \
        v = c_ok ? v + bias_t : 0.f;          /* (channels co .. co_fill - 1 of a 16-bit output are stored as zeros) */          \
        if (has_ct && c_ok) {                                                                                                    \
            const bool first = y == 0, last = y + 1 >= h_;                                                                       \
            const float t0 = first ? ct0a : (last ? ct0c : ct0b);                                                                \
            const float t1 = first ? ct1a : (last ? ct1c : ct1b);                                                                \
            const float t2 = first ? ct2a : (last ? ct2c : ct2b);                                                                \
            v += t0 + (float)xg * t1 + (float)y * t2;                                                                            \
        }                                                                                                                        \
        /* (one store instruction per wave and iteration whatever the masks: the count of the wait above is static) */           \
        const long long o = ((long long)y * wd_ + xg) * ldo_ + co_t;                                                             \
        const bool on = t_on && co_t < cofill_;                                                                                  \
        if (of32_) { if (on) ((float*)out_img)[o] = v; }                                                                         \
        else { if (on) st_from_float<T>((T*)out_img + o, v); }                                                                   \
    } while (0)
    for (int it3 = 0; it3 < iters; it3 += 3) {
        UPS_THIN_BODY(it3, 0, 1, 2);
        if (it3 + 1 < iters) UPS_THIN_BODY(it3 + 1, 1, 2, 0);
        if (it3 + 2 < iters) UPS_THIN_BODY(it3 + 2, 2, 0, 1);
    }
#undef UPS_THIN_BODY
#undef UPS_THIN_MMA
#undef UPS_THIN_LOAD_ROW
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef UPS_THIN_ISSUE_ROW
#undef UPS_THIN_DMA

template <typename T, int SW>
int launch_thinout(const ThinK& k, hipStream_t s) {
    constexpr int NR = SW == 16 ? 6 : 7;
    constexpr size_t smem = (size_t)NR * 8 * (SW + 2) * 64 + 2 * 8 * SW * 20 * 4;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_thinout_kernel<T, NR, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_thinout_kernel<T, NR, SW>), dim3(k.n * k.bands * k.strips), dim3(512), smem, s, k);
    return UPS_OK;
}

static int rows_on() {       // UPS_ROWS_KERNEL=0: these layers through the patch kernel (A/B runs); "force": also small batches (parity
    const char* e = getenv("UPS_ROWS_KERNEL");      // tests); read per call
    return (e && e[0] == '0') ? 0 : ((e && e[0] == 'f') ? 2 : 1);
}

template <typename T, int CI, int LW, int NR, bool FLIP, int DG>
int launch_rows(const RowsK& k, hipStream_t s) {
    constexpr int W = 1 << LW, KC = CI / 32, NCT = W / 16, NCG = 8 / NCT;
    constexpr size_t smem = (size_t)NR * KC * (W + 2) * 64 + (DG == 1 ? (size_t)(NR - 2) * NCG * W * 64 : (DG == 3 ? (size_t)(NR - 2) * 8 * 256 : 0)) +
                            8 * 16 * 80 + 256;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3x3_rows_kernel<T, CI, LW, NR, FLIP, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_rows_kernel<T, CI, LW, NR, FLIP, DG>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    return UPS_OK;
}

template <typename T>
int launch_rows_t(const RowsK& k, int ci, int w, bool flip, bool dg, hipStream_t s) {
    // ring depths: measured (tools/bench_conv.py ea_rb0 / ea_rb1): deeper rings at one block per CU lose to 8 rows at two blocks per CU
    // (0.42 vs 0.35 ms forward at 32 channels); 64 channels keep 144 weight registers per lane and run one block per CU either way
    // act' of the one-tile kernel: the producer's sign bytes (DG == 3) or the forward input's rows (DG == 1), both through the second
    // DMA ring
    if (ci == 32 && w == 128) {
        if (dg) {
            if (!flip) return 1;
            if (k.dact_bits) return launch_rows<T, 32, 7, 8, true, 3>(k, s);
            return launch_rows<T, 32, 7, 8, true, 1>(k, s);
        }
        return flip ? launch_rows<T, 32, 7, 8, true, 0>(k, s) : launch_rows<T, 32, 7, 8, false, 0>(k, s);
    }
    if (ci == 64 && w == 64) {
        if (dg) {
            if (!flip) return 1;
            if (k.dact_bits) return launch_rows<T, 64, 6, 10, true, 3>(k, s);
            return launch_rows<T, 64, 6, 10, true, 1>(k, s);
        }
        return flip ? launch_rows<T, 64, 6, 10, true, 0>(k, s) : launch_rows<T, 64, 6, 10, false, 0>(k, s);
    }
    return 1;
}

}  // namespace

// Internal entry used by ups_conv_igemm's dispatcher (conv_igemm.hip).  Returns 1 if the problem is not eligible, 0 after a launch,
// < 0 on a launch-setup failure.
int ups_conv3x3_rows_try(const ups_conv_desc* d, hipStream_t s) {
    if (!rows_on()) return 1;
    if (d->dtype != UPS_BF16 && d->dtype != UPS_F16) return 1;
    if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 1;
    if (d->hi != d->ho || d->wi != d->wo || d->out_h != d->ho || d->out_w != d->wo) return 1;
    const bool two = (d->ci == 64 && d->wi == 128) || (d->ci == 32 && d->wi == 256);      // two column tiles per wave (conv3x3_rows2_kernel)
    if (!((d->ci == 32 && d->wi == 128) || (d->ci == 64 && d->wi == 64) || two)) return 1;
    const int co_need = d->ci;
    if (d->co != co_need || d->co_fill != co_need || d->ldo < co_need || (d->ldo & 7) || (d->ldi & 7)) return 1;
    if (d->hi % 32 || d->hi < 32) return 1;
    if (rows_on() != 2 && (long long)d->n * (d->hi / 32) < (two ? 256 : 1024)) return 1;      // small batches: the patch kernel's 16x16 tiles fill the chip better
    if (d->act_in != UPS_ACT_NONE || d->coord_tab || d->mask_bits || d->mask_grad || d->d2s || d->f8_deq || d->in_f8 || d->out_f8 ||
        d->out_f8_amax || d->out_f32)
        return 1;
    if (d->res && !(d->res == d->in && d->ldr == d->ldi && d->ci == d->co)) return 1;
    bool fwd = true, flip = true;
    for (int t = 0; t < 9; ++t) {
        if (d->tap_w[t] != t) fwd = flip = false;
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) fwd = false;
        if (d->tap_dy[t] != 1 - t / 3 || d->tap_dx[t] != 1 - t % 3) flip = false;
    }
    if (!fwd && !flip) return 1;
    if (d->dact && (d->dtype != UPS_BF16 || !flip || (d->ldd & 7) || d->ldd < co_need || ((uintptr_t)d->dact & 15))) return 1;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15) || ((uintptr_t)d->w & 15)) return 1;      // 16-byte DMA pieces / stores
    if ((long long)d->hi * d->wi * (d->ldi > d->ldo ? d->ldi : d->ldo) * 2 >= (1ll << 31) || (long long)d->wi * d->ldi * 2 > 65536) return 1;
    RowsK k;
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.out = (unsigned char*)d->out;
    k.bias = d->bias; k.dact = (const unsigned char*)d->dact;
    // sign bytes: read by the one-tile input gradient (DG == 3: 4-byte DMA pieces, so ldd % 32 == 0), written by the one-tile forward
    k.dact_bits = (d->dact && !two && (d->ldd & 31) == 0 && ((uintptr_t)d->dact_bits & 3) == 0) ? (const unsigned char*)d->dact_bits : nullptr;
    k.sign_out = nullptr;                // (only the stride-2 row kernel writes sign bytes: ups_conv3x3_rows_s2_try)
    k.n = d->n; k.h = d->hi; k.ldi = d->ldi; k.ldo = d->ldo; k.ldd = d->ldd; k.co_tot = d->co;
    k.band_rows = 32; k.bands = d->hi / 32;
    k.res_self = d->res != nullptr; k.res_act = d->res_act; k.out_act = d->out_act;
    k.slope = d->act_slope; k.dact_ns = d->dact_kind == UPS_ACT_LRELU ? d->act_slope : 0.f;
    const bool dg = d->dact != nullptr;
    if (two && dg) return 1;        // the input gradient WITH act' of the two-tile shapes: the patch kernel (it reads the sign bytes)
    if (two) {
        if (d->ci == 64) {
            if (d->dtype == UPS_F16) return flip ? launch_rows2<f16, 64, 7, 8, true>(k, s) : launch_rows2<f16, 64, 7, 8, false>(k, s);
            return flip ? launch_rows2<bf16, 64, 7, 8, true>(k, s) : launch_rows2<bf16, 64, 7, 8, false>(k, s);
        }
        if (d->dtype == UPS_F16) return flip ? launch_rows2<f16, 32, 8, 8, true>(k, s) : launch_rows2<f16, 32, 8, 8, false>(k, s);
        return flip ? launch_rows2<bf16, 32, 8, 8, true>(k, s) : launch_rows2<bf16, 32, 8, 8, false>(k, s);
    }
    const int rc = d->dtype == UPS_F16 ? launch_rows_t<f16>(k, d->ci, d->wi, flip, dg, s) : launch_rows_t<bf16>(k, d->ci, d->wi, flip, dg, s);
    if (rc == 0 && k.sign_out) g_ups_sign_written = 1;
    return rc;
}

// Stride-2 forwards (even image sizes, taps at 2 y .. 2 y + 2): 32 -> 64 channels at 128x128, 64 -> 128 at 64x64.  Same return codes.
int ups_conv3x3_rows_s2_try(const ups_conv_desc* d, hipStream_t s) {
    if (!rows_on()) return 1;
    if (d->dtype != UPS_BF16 || d->ntaps != 9 || d->kh != 3 || d->kw != 3 || d->in_sy != 2 || d->in_sx != 2 || d->out_sy != 1 ||
        d->out_sx != 1 || d->out_oy || d->out_ox || d->out_h != d->ho || d->out_w != d->wo)
        return 1;
    const bool two = (d->ci == 32 && d->wi == 256 && d->co == 64) || (d->ci == 64 && d->wi == 128 && d->co == 128);     // the 256 x 256 configs
    if (!((d->ci == 32 && d->wi == 128 && d->co == 64) || (d->ci == 64 && d->wi == 64 && d->co == 128) || two)) return 1;
    if (d->hi != d->wi || d->ho * 2 != d->hi || d->wo * 2 != d->wi || d->co_fill != d->co || d->ldo < d->co || (d->ldo & 7) || (d->ldi & 7)) return 1;
    if (d->act_in != UPS_ACT_NONE || d->res || d->dact || d->coord_tab || d->d2s || d->out_f32 || d->mask_bits || d->mask_grad || d->f8_deq ||
        d->in_f8 || d->out_f8 || d->out_f8_amax || d->res_act)
        return 1;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 || d->tap_dx[t] != t % 3 || d->tap_w[t] != t) return 1;
    if (rows_on() != 2 && (long long)d->n * (d->ho / 32) < 512) return 1;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15) || ((uintptr_t)d->w & 15)) return 1;
    if ((long long)d->wi * d->ldi * 2 > 65536 || (long long)d->hi * d->wi * d->ldi * 2 >= (1ll << 31)) return 1;
    RowsK k;
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.out = (unsigned char*)d->out;
    k.bias = d->bias; k.dact = nullptr; k.dact_bits = nullptr; k.sign_out = two ? nullptr : (unsigned char*)d->sign_out;
    k.n = d->n; k.h = d->hi; k.ldi = d->ldi; k.ldo = d->ldo; k.ldd = 0; k.co_tot = d->co;
    k.band_rows = 32; k.bands = d->ho / 32;
    k.res_self = 0; k.res_act = 0; k.out_act = d->out_act;
    k.slope = d->act_slope; k.dact_ns = 0.f;
    if (two) return d->ci == 32 ? launch_rows_s2x2<32, 8, 7>(k, s) : launch_rows_s2x2<64, 7, 7>(k, s);
    const int rc = d->ci == 32 ? launch_rows_s2<32, 7, 7>(k, s) : launch_rows_s2<64, 6, 9>(k, s);
    if (rc == 0 && k.sign_out) g_ups_sign_written = 1;
    return rc;
}

// The logit convolution's forward: 256 input channels, <= 16 outputs, stride 1, no residual / activation; fp32 or 16-bit output.
int ups_conv3x3_thinout_try(const ups_conv_desc* d, hipStream_t s) {
    if (!rows_on()) return 1;
    if (d->dtype != UPS_BF16 && d->dtype != UPS_F16) return 1;
    if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 1;
    if (d->hi != d->ho || d->wi != d->wo || d->out_h != d->ho || d->out_w != d->wo) return 1;
    if (d->ci != 256 || d->co > 16 || d->co_fill > d->ldo || d->co_fill < d->co || (d->ldi & 7) || d->ldi < 256) return 1;
    if (d->out_f32 ? d->co_fill != d->co : ((d->ldo & 7) || d->co_fill > 16)) return 1;
    if (d->hi % 32 || d->wi % 32) return 1;
    if (d->act_in != UPS_ACT_NONE || d->res || d->dact || d->mask_bits || d->mask_grad || d->d2s || d->f8_deq || d->in_f8 || d->out_f8 ||
        d->out_f8_amax || d->out_act || d->res_act)
        return 1;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1 || d->tap_w[t] != t) return 1;
    const long long blocks = (long long)d->n * (d->hi / 32) * (d->wi / 32);
    if (rows_on() != 2 && blocks < 512) return 1;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->w & 15) || ((uintptr_t)d->out & (d->out_f32 ? 3 : 1))) return 1;
    if ((long long)d->wi * d->ldi * 2 + 1024 > 262144 || (long long)d->hi * d->wi * d->ldi * 2 >= (1ll << 31) || blocks >= (1ll << 31)) return 1;
    ThinK k;
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.out = (unsigned char*)d->out;
    k.bias = d->bias; k.coord_tab = d->coord_tab;
    k.n = d->n; k.h = d->hi; k.wd = d->wi; k.ldi = d->ldi; k.ldo = d->ldo; k.co = d->co; k.co_fill = d->co_fill; k.out_f32 = d->out_f32;
    k.strips = d->wi / 32; k.band_rows = 32; k.bands = d->hi / 32;
    // UPS_THIN_SW=16: 16-column strips, two blocks per CU (measured equal to the 32-column form at one block per CU: 0.43 ms both;
    // the 64-byte pieces at a 512-byte pixel pitch, not occupancy or prefetch depth, are what the launch is short of)
    const char* e = getenv("UPS_THIN_SW");
    if (e && e[0] == '1') {
        k.strips = d->wi / 16;
        return d->dtype == UPS_F16 ? launch_thinout<f16, 16>(k, s) : launch_thinout<bf16, 16>(k, s);
    }
    return d->dtype == UPS_F16 ? launch_thinout<f16, 32>(k, s) : launch_thinout<bf16, 32>(k, s);
}

// The part-masked first convolution's input gradient reduced to the hard mask (ups_conv_desc.mask_grad): 32 gradient channels,
// 128 columns, <= 4 view channels.
int ups_conv3x3_rows_maskgrad_try(const ups_conv_desc* d, hipStream_t s) {
    if (!rows_on()) return 1;
    if (!d->mask_grad || !d->mask_view || d->mask_bits || d->dtype != UPS_BF16) return 1;
    if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox) return 1;
    if (d->hi != d->ho || d->wi != d->wo || (d->wi != 128 && d->wi != 256) || d->hi % 32) return 1;
    if (d->ci != 32 || d->co != 3 || (d->ldi & 7) || d->ldi < 32) return 1;       // (three view channels: one dwordx3 load per pixel)
    if (d->act_in != UPS_ACT_NONE || d->res || d->dact || d->coord_tab || d->d2s || d->f8_deq || d->in_f8 || d->out_f8 || d->out_f8_amax ||
        d->out_act || d->res_act || d->bias)
        return 1;
    if (d->mask_batch <= 0 || d->n % d->mask_batch) return 1;
    for (int t = 0; t < 9; ++t)
        if (d->tap_dy[t] != 1 - t / 3 || d->tap_dx[t] != 1 - t % 3 || d->tap_w[t] != t) return 1;
    if (rows_on() != 2 && (long long)d->n * (d->hi / 32) < 1024) return 1;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->w & 15) || ((uintptr_t)d->mask_view & 3) || ((uintptr_t)d->mask_grad & 3)) return 1;
    if ((long long)d->wi * d->ldi * 2 > 262144 || (long long)d->hi * d->wi * d->ldi * 2 >= (1ll << 31)) return 1;
    MaskGK k;
    k.in = (const unsigned char*)d->in; k.w = (const unsigned char*)d->w; k.view = d->mask_view; k.g_hard = d->mask_grad;
    k.n = d->n; k.h = d->hi; k.ldi = d->ldi; k.co = d->co; k.B = d->mask_batch; k.P = d->n / d->mask_batch;
    k.band_rows = 32; k.bands = d->hi / 32;
    constexpr int NR = 8;
    if (d->wi == 128) {
        constexpr size_t smem = (size_t)NR * 130 * 64;
        static UpsPerDevice attr_set;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)conv3x3_rows_maskgrad_kernel<NR, 7, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return UPS_E_LAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL((conv3x3_rows_maskgrad_kernel<NR, 7, 1>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    } else {
        constexpr size_t smem = (size_t)NR * 258 * 64;
        static UpsPerDevice attr_set;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)conv3x3_rows_maskgrad_kernel<NR, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
                return UPS_E_LAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL((conv3x3_rows_maskgrad_kernel<NR, 8, 2>), dim3(k.n * k.bands), dim3(512), smem, s, k);
    }
    return UPS_OK;
}
