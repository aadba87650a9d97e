// Patch-tiled 3x3 / stride-1 'SAME' convolution (forward and dgrad) for the layers that carry >90 % of the
// FLOPs of the path (res-blocks of the mask decoder, the encoders and the perceptual trunk;
// cub/code/nn.py:617-664,1042-1056).
//
// Block = 512 threads = 8 waves, output tile = 16x16 pixels of one image x BN output channels.
// Per 64-byte channel chunk the (16+2)x(16+2) input halo patch is staged ONCE (activation-on-load and zero
// padding applied once per element instead of once per tap) and reused by all nine taps from LDS; the weights
// are staged one kernel row (3 taps) at a time.  Both are double-buffered, so the loop runs
//     issue global loads (next tap-row [+ next patch]) -> 24 MFMAs per wave from LDS -> write LDS -> 1 barrier.
// LDS pixel / weight rows are padded 64 -> 80 bytes (conflict-free ds_read_b128 for 16 consecutive pixels).
// MFMA: v_mfma_f32_32x32x16_bf16 or exact-fp32 v_mfma_f32_32x32x2_f32.  Epilogue identical to conv_igemm.
#include <stdlib.h>

#include "common.h"

#if defined(UPS_PHASE_TIMING)
// debug build (tools/probes/phase_timing.sh): thread 0 of each block records the 100 MHz wall clock at phase boundaries and the
// slot it ran in (HW_ID: CU / SE, XCC_ID, LDS base) -- where a block's life goes, which XCD ends when
__device__ unsigned long long ups_phase_t[8 * 65536];
extern "C" int ups_phase_dump(unsigned long long* host, int nblocks) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ups_phase_t), sizeof(unsigned long long) * 8 * (size_t)nblocks, 0, hipMemcpyDeviceToHost);
}
#define UPS_PHASE(k) do { if (threadIdx.x == 0 && blockIdx.x < 65536) ups_phase_t[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define UPS_PHASE(k)
#endif

namespace {

constexpr int TS = 16;                 // tile side
constexpr int PW = TS + 2;             // patch side
constexpr int PPIX = PW * PW;          // 324 patch pixels
constexpr int RS = 80;                 // LDS row stride of the register-staged fp32 weight tiles (bytes)
// Activation patch in LDS: [18 rows][PWP = 20 pixels][64 B], unpadded.  A pixel's four 16-byte chunks are XOR-swizzled
// with ((px >> 2) & 3): with a row pitch that is a multiple of 4 pixels this makes every ds_read_b128 fragment read
// conflict-free for the REAL b128 lane groups ({0-3,12-15,20-27}, ...: 8 pixels of tile row y and 8 of row y+1), for
// all nine tap shifts.  (The former 80-byte padded rows were 2-way conflicted on every A fragment read.)
constexpr int PWP = 20;
constexpr int APX = 64;                // bytes per patch pixel
constexpr int A_BYTES = PW * PWP * APX;   // 23040
__device__ __forceinline__ int a_swz(int px) { return (px >> 2) & 3; }
// bf16 path (v_mfma_f32_16x16x32_bf16 operands: lane l reads row l & 15, 16-byte chunk l >> 4): slot = chunk ^ a_swz16(px).
// Found by enumeration over the four ds_read_b128 lane groups: the only per-4-pixel-block XOR patterns that are
// conflict-free for a run of 16 pixels starting at patch column 0, 1 or 2 are (g, g^2, g, g^2, g) -- the same rule serves
// the weight rows (no shift).
__device__ __forceinline__ int a_swz16(int px) { return ((px >> 2) & 1) << 1; }
template <typename T> __device__ __forceinline__ int a_swz_t(int px) { return sizeof(T) == 2 ? a_swz16(px) : a_swz(px); }

template <int V> struct IC { static constexpr int value = V; };

struct PatchK {
    int n, h, w, ci, ldi, co, co_fill, ldo, ldr, ldd, act_in, out_f32, dact_kind, has_ctab;
    float act_slope;
    unsigned long long tap_off, tap_wi;   // 4 bits per tap: (dy+1)<<2|(dx+1) ; weight slice
    const void* in; const void* wgt; void* out;
    const float* bias; const float* coord_tab; const void* res; const void* dact;
    // bit-packed activation signs (ups_conv_desc.sign_out / dact_bits, ABI 4): [n][h][w][ld / 8] bytes
    unsigned char* sign_out; const unsigned char* dact_bits;
    // part-masked input / mask gradient (ups_conv_desc.mask_*): mask_B > 0 switches the block order to (image b, tile, part)
    // with the part fastest, so that the P blocks that read one view patch / write one g_hard line run back to back on one XCD
    const unsigned* mask; float* mask_grad; const float* mask_view;
    int mask_B, mask_P;
    // block index decode without integer divisions: n / d = umulhi(n, m) for n * d < 2^32 (launcher; m = 0: plain division)
    unsigned m_ntn, m_parts, m_tx, m_ty;
    // fp8 forward (ups_conv_desc.f8_*): wgt holds e4m3 weights scaled per output channel, f8_deq[c] = 1 / that scale,
    // *f8_scale the activation scale of this launch, f8_amax 64 slots that collect max |act(x)| for the next one
    const float* f8_deq; const float* f8_scale; float* f8_amax;
    int f8_e5m2;      // the staged tensor is a gradient: e5m2 operands (ups_conv_desc.f8_e5m2)
    // fp8 copies between layers (ups_conv_desc.in_f8 / out_f8_*): the consumer reads an already quantised tensor (16 channels per
    // 16-byte item, no conversion: the same staging registers as bf16, two blocks per CU), the producer's epilogue writes
    // act(out) * scale as e4m3 / e5m2 next to the bf16 tensor and records max |act(out)|
    const unsigned char* in_f8; unsigned char* out_f8; const float* out_f8_scale; float* out_f8_amax;
    int out_f8_act, out_f8_e5m2;
    // depth-to-space output (ups_conv_desc.d2s): GEMM channel ch = (py*2 + px) * (1 << d2s_shift) + c is channel c of output
    // pixel (2y + py, 2x + px) of a [n, 2h, 2w, ld] tensor (res / dact live on that lattice too)
    int d2s, d2s_shift;
    int taps_static;      // 1: forward order, 2: flipped (input-gradient) order, 0: neither (launcher)
    int out_act, res_act; // post-activation storage (ups_conv_desc.out_act / res_act)
    // the residual IS the input tensor (residual block, nn.py:1042-1056: res == in, same channel count): the 16x16 centre of
    // the halo patch of channel chunk c holds exactly the residual values of output channels 32c .. 32c+31, so each wave adds
    // its share into its accumulators straight from LDS while that chunk is resident -- the residual tensor is never read a
    // second time from global memory and the epilogue needs no residual tile (launcher sets it; 16-bit, SUB == TS)
    // res_patch == 2 (round 6): the same for the INPUT GRADIENT of a residual block, gx = act'(x) * conv^T(g) + g with act' from the producer's
    // sign bytes: the residual g is the launch's own operand, so it is added from the resident patch DIVIDED by act' (1 or 1 / slope:
    // the epilogue's multiplication by act' restores it) and the launch reads nothing but its operand, 32 bytes of signs per pixel and
    // the weights.  The tile's sign bytes sit in LDS from the prologue on (`sgn_off`: 256 pixels x BN / 8 bytes behind the buffers).
    int res_patch;
    int sgn_off;
    // sgn_res (round 6, late): EVERY one-tile-per-image 16-bit launch whose act' comes from sign bytes has the tile's bytes loaded by the
    // prologue (8 bytes per thread, under the first patch / weight round trip) instead of by the epilogue (one BYTE per lane and item,
    // a dependent round trip and a barrier in front of the accumulators: 6.9 of the 20.8 us of a block of the logit convolution's input
    // gradient).  Launcher: dact_bits given, ldd a multiple of 64, not depth-to-space, LDS permitting.
    int sgn_res;
};

__device__ __forceinline__ int fast_div(int n, int d, unsigned m) {
    if (d == 1) return n;
    return m ? (int)__umulhi((unsigned)n, m) : n / d;
}
static inline unsigned div_magic(long long n_max, int d) {
    if (d <= 1 || n_max * d >= (1ll << 32)) return 0u;
    return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d);
}

__device__ inline int p_dy(unsigned long long off, int t) { return (int)((off >> (4 * t + 2)) & 3) - 1; }
__device__ inline int p_dx(unsigned long long off, int t) { return (int)((off >> (4 * t)) & 3) - 1; }
__device__ inline int p_w(unsigned long long wi, int t) { return (int)((wi >> (4 * t)) & 15); }

template <typename T> struct PMma;
// bf16: three taps of one 32-channel chunk on v_mfma_f32_16x16x32_bf16 (one MFMA contracts the whole chunk: K = 32).
// A wave owns TM16 tile rows (16 pixels each) x TN16 groups of 16 output channels.  Same LDS bytes per FLOP as the
// 32x32x16 form (4 + 4 fragment reads per 16 MFMAs at TM16 = TN16 = 4) and the same cycles per FLOP, but the chip holds a
// higher clock on this shape (MI355X guide, DVFS give-back item 7; the dominant launch ran at ~1.7 GHz of 2.4).
// Fragment registers: the A fragments of a tap (TM16 x 4 VGPRs) and two slots of two B fragments; the next B pair is
// requested ahead of the current pair's MFMAs, the next tap's A fragments right behind the MFMAs that consumed the current
// ones (operands are read at issue).
typedef __attribute__((ext_vector_type(4))) float f32x4v;
// ROWB > 0: the wave's tile rows are ROWB bytes apart in the patch image (one image per tile): fragment i is read at an
// immediate offset i * ROWB from one per-tap lane address (no address arithmetic per read); ROWB == 0: rows from arow[].
typedef __attribute__((ext_vector_type(2))) long i64x2;
template <typename T, int F8> struct Frag16 { typedef i64x2 type; };
template <> struct Frag16<bf16, 0> { typedef bf16x8 type; };
template <> struct Frag16<f16, 0> { typedef f16x8 type; };
// F8: a 16-byte fragment holds 16 e4m3 channels -- two v_mfma_f32_16x16x32_fp8_fp8 per fragment pair (low / high 8 bytes:
// both operands use the same byte -> k mapping), a 64-byte pixel / weight row is a chunk of 64 channels.
// A2: the A fragments of the NEXT tap are requested at the start of the current one (a second register set: the instances
// whose patch arrives by LDS-DMA have the 16 VGPRs to spare) instead of behind its last MFMAs, where their LDS latency sat
// in front of every tap.
template <typename T, int TM16, int TN16, int ROWB, int F8, bool A2 = false>
__device__ __forceinline__ void bf16_taps16(const unsigned char* A, const unsigned char* B, const int (&arow)[TM16],
                                            int po0, int po1, int po2, int sw0, int sw1, int sw2, int b_tap_stride,
                                            f32x4v (&acc)[TM16][TN16]) {
    typedef typename Frag16<T, F8>::type frag_t;
    frag_t fa2[A2 ? 2 : 1][TM16], fb[2][2];
    auto fetch_a = [&](int t) __attribute__((always_inline)) {
        frag_t (&fa)[TM16] = fa2[A2 ? (t & 1) : 0];
        const int po = t == 0 ? po0 : (t == 1 ? po1 : po2);
        const int sw = t == 0 ? sw0 : (t == 1 ? sw1 : sw2);
        if constexpr (ROWB > 0) {
            const unsigned char* at = A + (arow[0] + po) + sw;       // uniform part added on the scalar unit, one v_add
#pragma unroll
            for (int i = 0; i < TM16; ++i) fa[i] = *(const frag_t*)(at + i * ROWB);
        } else {
#pragma unroll
            for (int i = 0; i < TM16; ++i) fa[i] = *(const frag_t*)(A + arow[i] + po + sw);
        }
    };
    auto fetch_b = [&](int t, int jh, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) fb[slot][jj] = *(const frag_t*)(B + t * b_tap_stride + (2 * jh + jj) * (16 * 64));
    };
    constexpr int NP = TN16 / 2;          // B fragment pairs per tap
    fetch_a(0);
    fetch_b(0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int jh = 0; jh < NP; ++jh) {
            const int step = t * NP + jh;
            // the next B pair is requested before this pair's MFMAs (second register slot), the next tap's A fragments behind
            // the last MFMAs that read the current ones
            if (step + 1 < 3 * NP) fetch_b((step + 1) / NP, (step + 1) % NP, (step + 1) & 1);
            if constexpr (A2) { if (jh == 0 && t + 1 < 3) fetch_a(t + 1); }
            frag_t (&fa)[TM16] = fa2[A2 ? (t & 1) : 0];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM16; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    // weights as the row operand: a lane then holds 4 CONSECUTIVE channels of one pixel (8-byte epilogue accesses)
                    if constexpr (F8 == 2) {        // e4m3 weights x e5m2 pixels (gradients)
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[step & 1][jj][0], fa[i][0], acc[i][2 * jh + jj], 0, 0, 0);
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[step & 1][jj][1], fa[i][1], acc[i][2 * jh + jj], 0, 0, 0);
                    } else if constexpr (F8 == 1) {
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[step & 1][jj][0], fa[i][0], acc[i][2 * jh + jj], 0, 0, 0);
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(fb[step & 1][jj][1], fa[i][1], acc[i][2 * jh + jj], 0, 0, 0);
                    } else if constexpr (sizeof(T) == 2 && !__is_same(T, bf16)) {      // fp16 forward tensors (mask decoder)
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[step & 1][jj], fa[i], acc[i][2 * jh + jj], 0, 0, 0);
                    } else {
                        acc[i][2 * jh + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[step & 1][jj], fa[i], acc[i][2 * jh + jj], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            if constexpr (!A2) { if (jh + 1 == NP && t + 1 < 3) fetch_a(t + 1); }
        }
    }
}

// fp8, block-scaled (F8 = 3 / 4): ONE tap of a 128-channel double chunk on v_mfma_scale_f32_16x16x128_f8f6f4 (E8M0 scales 1.0:
// the tensors carry per-tensor / per-channel scales applied in the epilogue).  The instruction takes 32 bytes per lane and
// operand -- lane (row l & 15, k-block l >> 4) holds k = 32 (l >> 4) .. + 31 of BOTH operands (tools/probes/mfma_scale_probe.hip)
// -- and the order of k is free as long as both operands agree: the lane's 32 bytes are the SAME 16-byte slot of two resident
// 64-byte chunk images (channels 16 q .. of chunk 2c and of chunk 2c + 1), so patch layout, swizzle and weight layout are those
// of the K = 32 path.  4x the K of the bf16 MFMA in 2x its cycles (measured 4.4 PFLOP/s against 1.5 for 16x16x32_fp8_fp8).
typedef __attribute__((ext_vector_type(8))) int i32x8v;
template <int TM16, int TN16, int ROWB, int ABYTES, int BHALF, bool E5M2, int BBUFS>
__device__ __forceinline__ void f8s_tap(const unsigned char* A, const unsigned char* B, int at, f32x4v (&acc)[TM16][TN16]) {
    typedef __attribute__((ext_vector_type(4))) int i32x4v;
    i32x8v fa[TM16], fb[BBUFS];      // (BBUFS = 1 at 64 accumulator registers: the second B slot does not fit 128 VGPRs)
    auto cat = [](i32x4v lo, i32x4v hi) __attribute__((always_inline)) -> i32x8v {
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto fetch_b = [&](int j, int slot) __attribute__((always_inline)) {
        fb[slot] = cat(*(const i32x4v*)(B + j * (16 * 64)), *(const i32x4v*)(B + BHALF + j * (16 * 64)));
    };
    fetch_b(0, 0);
#pragma unroll
    for (int i = 0; i < TM16; ++i) fa[i] = cat(*(const i32x4v*)(A + at + i * ROWB), *(const i32x4v*)(A + ABYTES + at + i * ROWB));
#pragma unroll
    for (int j = 0; j < TN16; ++j) {
        if (BBUFS == 2 && j + 1 < TN16) fetch_b(j + 1, (j + 1) & 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM16; ++i)      // weights as the row operand (e4m3), pixels e4m3 (activations) or e5m2 (gradients)
#if defined(UPS_ABLATE_MFMA)
            acc[i][j][0] += __int_as_float(fb[BBUFS == 2 ? (j & 1) : 0][i] ^ fa[i][j & 7]);
#else
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[BBUFS == 2 ? (j & 1) : 0], fa[i], acc[i][j], 0, E5M2 ? 1 : 0,
                                                                          0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#endif
        __builtin_amdgcn_s_setprio(0);
        // no hoisting of later fetches above these MFMAs (the scheduler otherwise requests every B fragment up front and
        // spills accumulators to hold them)
#pragma unroll
        for (int i = 0; i < TM16; ++i) asm volatile("" : "+v"(acc[i][j]));      // pins the MFMAs here (they are sunk to the loop end otherwise)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (BBUFS == 1 && j + 1 < TN16) fetch_b(j + 1, 0);
    }
}

template <> struct PMma<float> {
    template <int TM, int TN, int PITCH>
    __device__ static inline void tap(const unsigned char* a_pix, int hh, int v, const unsigned char* b_lane, f32x16 (&acc)[TM][TN]) {
        // a_pix = per-lane patch pixel base (chunk 0) shifted by the tap; lane half hh owns chunks 2hh, 2hh+1 (swizzled
        // by v); b_lane already includes the lane-half offset hh*32
        f32x4 a[TM][2], b[TN][2];
        const int c0 = ((2 * hh) ^ v) << 4, c1 = ((2 * hh + 1) ^ v) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a[i][0] = *(const f32x4*)(a_pix + i * (2 * PITCH * APX) + c0);
            a[i][1] = *(const f32x4*)(a_pix + i * (2 * PITCH * APX) + c1);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            b[j][0] = *(const f32x4*)(b_lane + j * (32 * RS));
            b[j][1] = *(const f32x4*)(b_lane + j * (32 * RS) + 16);
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

// OCC = 1: one block per CU (3-stage weight ring, double-buffered patch, <= 256 VGPRs);
// OCC = 2 (bf16 only): two blocks per CU (2-stage ring, single patch buffer, activation-derivative tile staged as sign
// bytes, <= 80 KB LDS and <= 128 VGPRs) -- the second block's MFMAs hide this block's prologue, patch re-staging and
// epilogue, none of which overlap anything when a CU holds a single block.
// SUB = 16: a tile is a 16x16 window of one image (halo from the neighbouring pixels).  SUB = 8 / 4: the images themselves
// are 8x8 / 4x4 (encoder bottoms, first decoder levels, VGG block 5) and a tile packs G x G = 4 / 16 whole images, each
// with its own all-zero halo (the patch grid is G*(SUB+2) wide; halo slots are zeroed once and never written).
// TAPS = 1 / 2: the nine taps are in the forward's r-major order (dy = t/3 - 1, dx = t%3 - 1, weight slice t) / in the input
// gradient's flipped order (dy = 1 - t/3, dx = 1 - t%3, weight slice t), known at compile time: the tap-row loop of the
// two-blocks-per-CU variant is then unrolled over (tap-row, ring stage) with every LDS offset an immediate and no tap decode
// on the scalar unit (the generic loop spent ~200 SALU + 18 VALU per 48 MFMAs on it).  TAPS = 0: taps from the descriptor.
// DMAP: the input needs nothing done to it on the way (act_in none, no part mask, bf16 / fp16, whole 32-channel chunks): the
// halo patch then arrives by LDS-DMA like the weights -- 23 wave-instructions of 1 KiB per chunk, the swizzle applied on the
// source side, patch pixels outside the image masked off in EXEC (their slots were zeroed once and are never written) -- with
// no staging registers, no VALU and no ds_write.  Every input-gradient launch qualifies (its input is the gradient tensor).
// CSTD: chunk-granular weight stages (CHUNKST below) for a one-tile-per-image instance with the DMA patch -- chosen by the launcher
// for grids of at most one 128-wide block per CU (16x16 / 32x32 maps at small batches), as 64-wide tiles: twice the blocks, one
// barrier and one exposed L2 round trip per 32-channel chunk instead of three
template <typename T, int BN, int OCC, int SUB, int F8 = 0, bool PRE = false, int TAPS = 0, bool DMAP = false, bool CSTD = false>
__global__ __launch_bounds__(512, 2 * OCC) void conv3x3_patch_kernel(const PatchK p, const int tiles_x, const int tiles_y,
                                                                 const int ntn, const int kchunks, const int nblocks) {
    constexpr int EPC = Chunk<T>::N;
#if defined(UPS_PATCH_A2)
    constexpr bool A2FR = DMAP;                  // second A-fragment register set (bf16_taps16): measured SLOWER (round 3: the 128-wide
#else                                            // two-blocks-per-CU instances spill at 128 VGPRs: dgrad 2.28 -> 3.07 ms), off by default
    constexpr bool A2FR = false;
#endif
    constexpr int BK = 4 * EPC;                  // weight-row elements of T per 64-byte row (the fp8 rows are addressed as T too)
    constexpr int BKA = F8 ? 64 : BK;            // input channels per chunk
    constexpr bool F8S = F8 >= 3;                // block-scaled K = 128 MFMA over pairs of 64-channel chunks (f8s_tap)
    constexpr int F8K = F8S ? F8 - 2 : F8;       // 1: e4m3 pixels (activations), 2: e5m2 pixels (gradients)
    static_assert(!F8 || (sizeof(T) == 2 && SUB == TS && (OCC == 1 || PRE)), "fp8 operands: bf16 tensors, one tile per image; two blocks per CU only with a pre-quantised input");
    static_assert(!F8S || (PRE && OCC == 2 && BN >= 64 && TAPS == 0 && !DMAP), "block-scaled fp8: pre-quantised input, two blocks per CU");
    static_assert(!PRE || F8, "a pre-quantised input implies fp8 operands");
#if defined(UPS_F8S_WN1)
    // (block-scaled fp8, alternative wave tile: 2 tile rows x ALL the N-tile's channels -- 16 A registers resident, the B fragments
    // stream through two slots; 20 instead of 16 fragment reads per step and an epilogue that spills: not the default)
    constexpr int WN = (BN == 32 || F8 >= 3) ? 1 : 2;
#else
    constexpr int WN = (BN == 32) ? 1 : 2;
#endif
    constexpr int WM = 8 / WN;                   // 4 or 8 waves along the pixels
    constexpr int TM = 256 / WM / 32;            // 2 or 1
    constexpr int TN = BN / WN / 32;             // 2, 1, 1
    constexpr int B_BYTES = 3 * BN * RS;         // one tap-row of weights
    constexpr int NB = (3 * BN * 4 + 511) / 512; // weight chunks per thread per tap-row: 3, 2, 1
    constexpr int HALF_OFF = (sizeof(T) == 2) ? 16 : 32;
    constexpr int G = TS / SUB;                  // sub-images per tile side
    constexpr int PR = G * (SUB + 2);            // patch rows = columns: 18 / 20 / 24
    constexpr int PWPS = (PR + 3) / 4 * 4;       // row pitch in pixels (multiple of 4): 20 / 20 / 24
    constexpr int ABY = PR * PWPS * APX;         // bytes of one patch buffer
    static_assert(SUB == 16 || OCC == 1, "multi-image tiles run one block per CU");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Abuf = smem;                  // 2 x ABY
    unsigned char* Bbuf = smem + 2 * ABY;        // 2 x B_BYTES

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values derived from it live in SGPRs
    UPS_PHASE(0);
#if defined(UPS_PHASE_TIMING)
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        const unsigned lb = __builtin_amdgcn_s_getreg((7 << 11) | 6);
        ups_phase_t[blockIdx.x * 8 + 7] = (unsigned long long)hw | ((unsigned long long)xcc << 32) | ((unsigned long long)lb << 40);
    }
#endif
    const float act_ns = ups_slope_eff(p.act_in, p.act_slope);   // branch-free activation-on-load
    const float dact_ns = ups_slope_eff(p.dact_kind, p.act_slope); // act'(x) = x > 0 ? 1 : dact_ns (only used when dact != NULL)
    const float oact_ns = ups_slope_eff(p.out_act, p.act_slope);   // stored value = max(v, oact_ns * v) when out_act is set
    const float res_inv = p.res_act ? 1.f / p.act_slope : 1.f;     // residual stored as leaky-ReLU(x): x = r > 0 ? r : r / slope
    // XCD-aware order: consecutive logical tiles (which share halos / the same patch for both N-tiles) stay on
    // one XCD's L2 (blocks are dealt round-robin over the 8 XCDs)
    int bid = blockIdx.x;
    if ((nblocks & 7) == 0) bid = (bid & 7) * (nblocks >> 3) + (bid >> 3);
    int t = fast_div(bid, ntn, p.m_ntn);
    const int nt = bid - t * ntn;
    int part = 0;
    if (p.mask_B > 0) { const int q = fast_div(t, p.mask_P, p.m_parts); part = t - q * p.mask_P; t = q; }
    const int t1 = fast_div(t, tiles_x, p.m_tx), t2 = fast_div(t1, tiles_y, p.m_ty);
    const int tx0 = (t - t1 * tiles_x) * TS;
    const int ty0 = (t1 - t2 * tiles_y) * TS;
    const int img = t2 * (G * G);                // first image of the tile (part mode: the view image b)
    const int img_pm = p.mask_B > 0 ? part * p.mask_B + img : img;      // part-major image p * B + b
    const int img_in = p.mask ? img : img_pm;    // masked forward reads the view; the mask-gradient pass reads d(out) of part image
    const int wm = wid / WN, wn = wid % WN;

    const T* __restrict__ in = (const T*)p.in + (long long)img_in * p.h * p.w * p.ldi;
    const unsigned* __restrict__ mbits = p.mask ? p.mask + (long long)img * p.h * p.w : nullptr;
    const T* __restrict__ w = (const T*)p.wgt;

    // ---- patch staging: items tid, tid+512, tid+1024 of the 324x4 16-byte chunks
    // One tile per image (SUB == TS): ONE dword per item held across the channel loop -- low 16 bits = pixel index relative
    // to the patch origin in IMAGE pitch (py * w + px; 0xffff = outside the image, zero fill), high 16 bits = LDS byte offset
    // of the slot (swizzled patch layout).  The load address is a uniform base (image + chunk + patch origin, scalar) plus
    // rel * row bytes: one v_mad per item and chunk instead of the divisions / bounds tests (the launcher checks
    // 18 * w < 65535).  Multi-image tiles keep separate 32-bit offsets (one block per CU there, registers are free).
    unsigned pk0 = 0xffffu, pk1 = 0xffffu, pk2 = 0xffffu;
    int pa0 = -1, pa1 = -1, pa2 = -2;
    int sa0 = 0, sa1 = 0, sa2 = 0;
    const int origin = (ty0 - 1) * p.w + (tx0 - 1);      // may be negative on the top / left edge; valid items never are
    const bool has2 = tid + 1024 < PPIX * 4;
    if constexpr (SUB == TS) {
        auto mk = [&](int item) -> unsigned {
            item = min(item, PPIX * 4 - 1);
            const int pix = item >> 2, ch = item & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int y = ty0 - 1 + py, x = tx0 - 1 + px;
            const unsigned sa = (py * PWPS + px) * APX + ((ch ^ a_swz_t<T>(px)) << 4);
            const unsigned rel = ((unsigned)y >= (unsigned)p.h || (unsigned)x >= (unsigned)p.w) ? 0xffffu : (unsigned)(py * p.w + px);
            return (sa << 16) | rel;
        };
        pk0 = mk(tid); pk1 = mk(tid + 512); pk2 = mk(tid + 1024);
    } else {
        // only the 256 interior pixels are ever loaded (2 items per thread); every halo slot stays zero
        auto mk = [&](int item, int& sa) -> int {
            const int q = item >> 2, ch = item & 3;
            const int ty = q >> 4, tx = q & 15;
            const int sy = ty / SUB, ly = ty - sy * SUB, sx = tx / SUB, lx = tx - sx * SUB;
            const int px = tx + 2 * sx + 1;
            sa = ((ty + 2 * sy + 1) * PWPS + px) * APX + ((ch ^ a_swz_t<T>(px)) << 4);
            return ((((sy * G + sx) * SUB + ly) * SUB + lx) * p.ldi + ch * EPC) * (int)sizeof(T);
        };
        pa0 = mk(tid, sa0); pa1 = mk(tid + 512, sa1);
        for (int i = tid * 16; i < 2 * ABY; i += 512 * 16) *(uint4*)(Abuf + i) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    // DMAP: piece j = wid + 8 q fills LDS bytes [j * 1024, +1024) of the patch image = pixel slots 16 j .. 16 j + 15
    unsigned pd_off[3] = {0u, 0u, 0u};
    unsigned long long pd_mask[3] = {0ull, 0ull, 0ull};
    if constexpr (DMAP || F8S) {
        // (F8S: the pre-quantised tensor has one byte per channel -- a pixel's 64-byte chunk is the same four 16-byte slots)
        static_assert(SUB == TS && (F8 == 0 || F8S) && sizeof(T) == 2, "DMA patch: one tile per image, 16-bit tensors or an fp8 copy");
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int pp = (wid + 8 * q) * 16 + (lane >> 2), sl = lane & 3;
            const int py = pp / PWPS, px = pp - py * PWPS;
            const int y = ty0 - 1 + py, x = tx0 - 1 + px;
            const bool ok = py < PW && px < PW && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
            pd_off[q] = (unsigned)(py * p.w + px) * ((unsigned)p.ldi * (F8S ? 1u : 2u)) + (unsigned)((sl ^ a_swz16(px)) << 4);
            pd_mask[q] = __ballot(ok);
        }
        // halo slots outside the image are never written by the DMA: zero the patch buffer(s) once, on border tiles only
        if (ty0 == 0 || tx0 == 0 || ty0 + TS >= p.h || tx0 + TS >= p.w) {
            const int nab = F8S ? 2 : ((kchunks == 1 || OCC == 2) ? 1 : 2);
            for (int i = tid * 16; i < nab * ABY; i += 512 * 16) *(uint4*)(Abuf + i) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
    }
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    uint4 ra0, ra1, ra2;
    uint4 rb0, rb1, rb2;        // fp8: channels 8..15 of the items (an item is 16 bf16 channels -> 16 e4m3 bytes)
    float f8_amax_t = 0.f;      // fp8: max |act(x)| this thread has staged
    struct WSet { uint4 r0, r1, r2; } ws0, ws1;     // two weight stages in flight (prefetch distance 2)

    const int cha = (tid & 3) * (F8 ? 16 : EPC); // 512 % 4 == 0: all three items of a thread share the chunk slot
    // branch-free: always load (from the tensor base when masked) and select, so the number of loads in flight
    // is static and the compiler can use counted vmcnt waits across the prefetch distance
    auto ld_a = [&](int off, int koff) -> uint4 {
        uint4 v = zero4;
#if !defined(UPS_ABLATE_GLOAD)
        if (off >= 0 && koff + cha < p.ci) {
            int o = off;
            asm volatile("" : "+v"(o));     // keep the 32-bit offset: scalar base (image + chunk) + vector offset addressing
            v = *(const uint4*)((const unsigned char*)(in + koff) + (unsigned)o);
        }
#endif
        return v;
    };
    const T* __restrict__ in_o = in + (long long)origin * p.ldi;
    const unsigned* __restrict__ mbits_o = mbits ? mbits + origin : nullptr;
    const unsigned row_b = (unsigned)p.ldi * (unsigned)sizeof(T);
    const unsigned char* __restrict__ in8_o = PRE ? p.in_f8 + ((long long)img_in * p.h * p.w + origin) * p.ldi : nullptr;
    auto ld_rel = [&](unsigned pk, int koff, uint4& hi) -> uint4 {
        uint4 v = zero4;
        if constexpr (F8) hi = zero4;
#if !defined(UPS_ABLATE_GLOAD)
        asm volatile("" : "+v"(pk));        // derived offsets are recomputed per chunk, not hoisted into held registers
        const unsigned rel = pk & 0xffffu;
        if (rel != 0xffffu && koff + cha < p.ci) {
            // scalar base (image + chunk + origin) + 32-bit vector offset addressing
            if constexpr (PRE) {        // pre-quantised tensor: one byte per channel, the item's 16 channels are 16 bytes
                const unsigned o8 = __umul24(rel, (unsigned)p.ldi) + (unsigned)cha;
                v = *(const uint4*)(in8_o + koff + o8);
            } else {
                const unsigned o = __umul24(rel, row_b) + (unsigned)(cha * (int)sizeof(T));
                v = *(const uint4*)((const unsigned char*)(in_o + koff) + o);
                if constexpr (F8) hi = *(const uint4*)((const unsigned char*)(in_o + koff) + o + 16);
            }
            // part-masked input (model.py:185): the pixel belongs to this block's part image only where its hard-mask bit is set
            if (mbits_o && !((mbits_o[rel] >> part) & 1u)) v = zero4;
        }
#endif
        return v;
    };
    auto load_patch = [&](int cc) __attribute__((always_inline)) {
        const int koff = cc * BKA;
        if constexpr (SUB == TS) {
            ra0 = ld_rel(pk0, koff, rb0); ra1 = ld_rel(pk1, koff, rb1);
            if (has2) ra2 = ld_rel(pk2, koff, rb2);
        } else {
            ra0 = ld_a(pa0, koff); ra1 = ld_a(pa1, koff);
        }
    };
    auto act_u4 = [&](uint4 u) -> uint4 {
        if (p.act_in != UPS_ACT_NONE) {
            u = ups_act_chunk<(OCC == 2 && BN == 128)>(u, act_ns, (T*)nullptr);
        }
        return u;
    };
    // fp8: 16 bf16 channels -> activation -> running max -> * scale -> 16 e4m3 bytes (v_cvt_pk_fp8_f32: RNE, saturating)
    float f8_sa = 1.f;
    if constexpr (F8) f8_sa = *p.f8_scale;
    auto cvt_f8 = [&](uint4 lo, uint4 hi) __attribute__((always_inline)) -> uint4 {
        const unsigned wsrc[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned wd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[4] = {__uint_as_float(wsrc[2 * k] << 16), __uint_as_float(wsrc[2 * k] & 0xffff0000u),
                          __uint_as_float(wsrc[2 * k + 1] << 16), __uint_as_float(wsrc[2 * k + 1] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act_in != UPS_ACT_NONE) f[e] = ups_act_ns(f[e], act_ns);
                f8_amax_t = fmaxf(f8_amax_t, fabsf(f[e]));
                constexpr float FMAX = F8K == 2 ? 57344.f : 448.f;      // e5m2 / e4m3 largest normal
                f[e] = __builtin_amdgcn_fmed3f(f[e] * f8_sa, -FMAX, FMAX);
            }
            int d = 0;
            if constexpr (F8K == 2) {
                d = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], d, false);
                d = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], d, true);
            } else {
                d = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], d, false);
                d = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], d, true);
            }
            wd[k] = (unsigned)d;
        }
        return make_uint4(wd[0], wd[1], wd[2], wd[3]);
    };
    auto store_patch = [&](unsigned char* A) __attribute__((always_inline)) {
        if constexpr (PRE) {
            *(uint4*)(A + (pk0 >> 16)) = ra0;
            *(uint4*)(A + (pk1 >> 16)) = ra1;
            if (has2) *(uint4*)(A + (pk2 >> 16)) = ra2;
        } else if constexpr (F8) {
            *(uint4*)(A + (pk0 >> 16)) = cvt_f8(ra0, rb0);
            *(uint4*)(A + (pk1 >> 16)) = cvt_f8(ra1, rb1);
            if (has2) *(uint4*)(A + (pk2 >> 16)) = cvt_f8(ra2, rb2);
        } else if constexpr (SUB == TS) {
            // one item at a time (the scheduler would otherwise unpack all three chunks at once: 24 more live registers)
            auto slot = [&](unsigned pk) __attribute__((always_inline)) -> unsigned { asm volatile("" : "+v"(pk)); return pk >> 16; };
            *(uint4*)(A + slot(pk0)) = act_u4(ra0);
            __builtin_amdgcn_sched_barrier(0);
            *(uint4*)(A + slot(pk1)) = act_u4(ra1);
            __builtin_amdgcn_sched_barrier(0);
            if (has2) *(uint4*)(A + slot(pk2)) = act_u4(ra2);
        } else {
            *(uint4*)(A + sa0) = act_u4(ra0);
            *(uint4*)(A + sa1) = act_u4(ra1);
        }
    };
    // weights of tap-row g (taps 3g..3g+2), chunk cc: item -> (tap_local, row, ch)
    // weights are stored blocked-K: [tap][k-chunk][row][BK] (conv_aux.hip) -> a tile is one contiguous range
    auto ld_b = [&](int item, int g, int cc) -> uint4 {
        uint4 v = zero4;
        if (item < 3 * BN * 4) {
            const int tl = item / (BN * 4), rem = item - tl * (BN * 4);
            const int row = rem >> 2, ch = rem & 3;
            const int c = nt * BN + row;
#if !defined(UPS_ABLATE_GLOAD)
            if (c < p.co)
                v = *(const uint4*)(w + (((long long)p_w(p.tap_wi, 3 * g + tl) * kchunks + cc) * p.co + c) * BK + ch * EPC);
#endif
        }
        return v;
    };
    auto load_w = [&](WSet& q, int g, int cc) {
        q.r0 = ld_b(tid, g, cc);
        if (NB > 1) q.r1 = ld_b(tid + 512, g, cc);
        if (NB > 2) q.r2 = ld_b(tid + 1024, g, cc);
    };
    auto store_w = [&](const WSet& q, unsigned char* B) {
        if (tid < 3 * BN * 4) *(uint4*)(B + (tid >> 2) * RS + (tid & 3) * 16) = q.r0;
        if (NB > 1 && tid + 512 < 3 * BN * 4) *(uint4*)(B + ((tid + 512) >> 2) * RS + (tid & 3) * 16) = q.r1;
        if (NB > 2 && tid + 1024 < 3 * BN * 4) *(uint4*)(B + ((tid + 1024) >> 2) * RS + (tid & 3) * 16) = q.r2;
    };

    // fp32 path: 32x32 accumulator blocks (exact-f32 MFMA 32x32x2); bf16 path: 16x16 blocks (MFMA 16x16x32), TM16 tile rows x
    // TN16 groups of 16 channels per wave.  Only the set of the instantiated dtype is live.
    constexpr int TM16 = 2 * TM, TN16 = 2 * TN;
    f32x16 acc[TM][TN];
    f32x4v acc16[TM16][TN16];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < TM16; ++i)
#pragma unroll
            for (int j = 0; j < TN16; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    // per-lane fragment bases: output row r of MFMA block tm -> tile pixel (ty, tx)
    const int r = lane & 31, hh = lane >> 5;
    const int ty_l = (wm * TM * 2) + (r >> 4), tx_l = r & 15;
    // patch coordinates of the lane's pixel minus the halo origin: tap (dy,dx) adds (dy+1, dx+1)
    const int py_l = ty_l + 2 * (ty_l / SUB) * (SUB < TS ? 1 : 0), px_l = tx_l + 2 * (tx_l / SUB) * (SUB < TS ? 1 : 0);
    const int a_lane_off = (py_l * PWPS + px_l) * APX;
    const int b_lane_off = (wn * TN * 32 + r) * RS + hh * HALF_OFF;
    // bf16 (16x16x32 operands): lane = (pixel p16 of a 16-pixel tile row, 16-byte chunk q16); fragment i = tile row wm*TM16+i
    const int p16 = lane & 15, q16 = lane >> 4;
    const int px_l16 = p16 + 2 * (p16 / SUB) * (SUB < TS ? 1 : 0);
    const int a_lane16 = px_l16 * APX;
    int arow16[TM16];                 // wave-uniform row offsets (SGPRs); the lane part travels with the swizzle term
#pragma unroll
    for (int i = 0; i < TM16; ++i) {
        const int ty = wm * TM16 + i;
        arow16[i] = (ty + 2 * (ty / SUB) * (SUB < TS ? 1 : 0)) * PWPS * APX;
    }

    const int total = 3 * kchunks;
    if constexpr (sizeof(T) == 2) {
        // ===== bf16: weights by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), three stages, counted vmcnt.
        // A stage is [3 taps x BN rows][64 B] unpadded (the DMA writes 64 lanes x 16 B contiguously); bank conflicts of
        // the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE side: LDS slot
        // (row, s) holds chunk s ^ a_swz16(row).  The DMA of tap-row it+2 is issued before the MFMAs of tap-row it
        // and only waited for (s_waitcnt vmcnt(NW), raw s_barrier) at the end of tap-row it+1.
        constexpr int BST = 3 * BN * 64;                 // bytes per weight stage
        constexpr int NJ = 3 * BN * 4 / 64;              // DMA wave-instructions per stage: 24 / 12 / 6
        constexpr int NW = (NJ + 7) / 8;                 // per wave: 3 / 2 / 1
        // CHUNKST (multi-image tiles, N-tiles up to 64 wide): the 4x4 / 8x8 maps run as grids of 64-160 blocks whose 24 tap-rows of
        // 4-8 MFMAs per wave sit behind a weight stream two tap-rows deep -- one L2 round trip per two tap-rows and nothing else
        // (25-48 us per launch for 2-10 us of MFMA work).  There a stage is a whole 32-channel chunk (all nine taps: 3 x BST), two
        // of them: the next chunk's 18 / 37 KB are requested at once, one barrier and one exposed round trip per CHUNK
        constexpr bool CHUNKST = (SUB < TS || (CSTD && DMAP && SUB == TS)) && BN <= 64 && F8 == 0 && OCC == 1;
        constexpr int NST = (OCC == 2) ? 2 : (CHUNKST ? 6 : 3);          // ring stages (tap-rows)
#if defined(UPS_OCC2_FRAG2)
        constexpr int FRAG_BUFS = 2;
#else
        constexpr int FRAG_BUFS = (BN == 128) ? 1 : 2;   // 128-wide tiles at two blocks per CU: <= 128 VGPRs without spills
#endif
        // a single-chunk problem (ci <= 32) never touches the second patch buffer: the launcher then requests less LDS
        // (2 blocks per CU instead of 1, which hides the per-block load latency of these 3-iteration blocks)
        unsigned char* Bst = smem + (((kchunks == 1 && SUB == TS) || OCC == 2) ? 1 : 2) * ABY;   // NST x BST
        // B fragment of channel group j: row (wn*TN16 + j)*16 + p16 of the stage, chunk q16 under the source-side swizzle
        const int boff16 = p16 * 64 + ((q16 ^ a_swz16(p16)) << 4);
        // per-lane source decode of this wave's DMA instructions (loop invariant)
        int d_tl[NW];
        unsigned d_off[NW];                              // byte offset of the lane's 16 bytes inside one (tap, k-chunk) slab
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int j = (wid + 8 * q) % NJ;
            const int pos = j * 64 + lane, row = pos >> 2, slot = pos & 3;
            const int tl = (j * 16) / BN, rl = row - tl * BN;   // a wave-instruction's 16 rows lie in one tap (wave-uniform)
            const int c = min(nt * BN + rl, p.co - 1);   // rows past co: duplicate a valid row (masked in the epilogue)
            d_tl[q] = tl;
            d_off[q] = (unsigned)(c * BK + (slot ^ a_swz16(row)) * 8) * 2u;
        }
        const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
        auto dma_w = [&](int it) __attribute__((always_inline)) {
            const int cc = it / 3, g = it - cc * 3;
            unsigned char* stg = Bst + (it % NST) * BST;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const int j = (wid + 8 * q) % NJ;
                // uniform slab base (scalar) + per-lane 32-bit offset
                const int wslice = TAPS != 0 ? 3 * g + d_tl[q] : p_w(p.tap_wi, 3 * g + d_tl[q]);
                const T* slab = w + ((long long)wslice * kchunks + cc) * p.co * BK;
#if !defined(UPS_ABLATE_DMA)
                // issued as inline asm: hipcc's wait-count pass treats the builtin as a FLAT access that may touch LDS
                // and from then on waits lgkmcnt(0) before every fragment use (no counted waits); hidden from the pass,
                // the fragment waits are counted.  The pass's own vmcnt(N) waits stay safe (extra younger operations
                // only make them wait longer).
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(stg + j * 1024 - smem));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(lds_dst), "v"(d_off[q]), "s"(slab) : "memory", "m0");
#endif
            }
        };
        const unsigned char* __restrict__ in_b = (const unsigned char*)(in + (long long)origin * p.ldi);
        auto dma_patch = [&](int cc, int boff = 0) __attribute__((always_inline)) {
            if constexpr (DMAP) {
                const unsigned char* base = in_b + cc * 64;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (wid + 8 * q >= (PW * PWPS + 15) / 16) continue;        // piece 23 of 22.5 (wave-uniform)
                    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(boff + (wid + 8 * q) * 1024));
#if !defined(UPS_ABLATE_GLOAD)
                    asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                                 :: "s"(pd_mask[q]), "s"(lds_dst), "v"(pd_off[q]), "s"(base) : "memory", "m0");
#endif
                }
            }
        };
        // lane part of the A fragment address for the three column shifts (static taps: loop invariant)
        const int swx0 = a_lane16 + ((q16 ^ a_swz16(px_l16 + 0)) << 4);
        const int swx1 = a_lane16 + ((q16 ^ a_swz16(px_l16 + 1)) << 4);
        const int swx2 = a_lane16 + ((q16 ^ a_swz16(px_l16 + 2)) << 4);
        // residual from the resident patch (PatchK.res_patch): chunk cc = the output channels 32 (cc - c_first) .. +31 of this
        // N-tile; wave wn owns CPW = TN16 / 2 of the tile's chunks; lane (p16, q16) adds the 4 channels 4 q16 .. of accumulator
        // block (i, j) = pixel (tile row wm*TM16 + i, column p16): 8 bytes at patch pixel (row + 1, p16 + 1)
        constexpr int CPW = TN16 / 2 > 0 ? TN16 / 2 : 1;
        const int rp_first = nt * (BN / 32) + wn * CPW;
        const int rp_lane = ((px_l16 + 1) * APX) + ((((q16 >> 1)) ^ a_swz16(px_l16 + 1)) << 4) + (q16 & 1) * 8;   // jj = 0; jj = 1: slot ^ 2
        // (flipped-tap instances, res_patch == 2: the residual goes in divided by act' -- the lane's four sign bits of pixel (row, p16),
        // channels 32 chunk + 16 jj + 4 q16 .. + 3, from the tile's resident sign bytes)
        constexpr bool RP2 = DMAP && SUB == TS && TN16 >= 2 && TAPS == 2 && BN == 128 && F8 == 0 && __is_same(T, bf16);
        const unsigned char* S1 = smem + p.sgn_off;
        const float rp2_inv = (RP2 && p.res_patch == 2) ? 1.f / dact_ns : 1.f;
        auto add_res_patch = [&](const unsigned char* A, int cc) __attribute__((always_inline)) {
            if constexpr (DMAP && SUB == TS && TN16 >= 2) {     // (the conditions of res_patch are those of the DMA patch)
                const int c2 = cc - rp_first;
                if (p.res_patch && c2 >= 0 && c2 < CPW) {
                    // (the lane part of the sign-byte address is re-derived here, once per channel chunk, from a copy of the lane id the
                    // compiler cannot hoist: two more loop-invariant registers would push the 128-register instances into scratch)
                    int ln = lane;
                    if constexpr (RP2) asm volatile("" : "+v"(ln));
                    const int sg_lane = (ln & 15) * (BN / 8) + (ln >> 5), sg_sh = (ln >> 2) & 4;
#pragma unroll
                    for (int jh = 0; jh < CPW; ++jh) {
                        if (jh != c2) continue;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                            for (int i = 0; i < TM16; ++i) {
                                const uint2 rv = *(const uint2*)(A + (wm * TM16 + i + 1) * (PWPS * APX) + (rp_lane ^ (jj << 5)));
                                float r0, r1, r2, r3;
                                ups_unpack2<T>(rv.x, r0, r1); ups_unpack2<T>(rv.y, r2, r3);
                                if constexpr (RP2) {
                                    if (p.res_patch == 2) {
                                        const unsigned nib = (unsigned)S1[(wm * TM16 + i) * 16 * (BN / 8) + 4 * (wn * CPW + jh) + 2 * jj + sg_lane] >> sg_sh;
                                        r0 = (nib & 1u) ? r0 : r0 * rp2_inv; r1 = (nib & 2u) ? r1 : r1 * rp2_inv;
                                        r2 = (nib & 4u) ? r2 : r2 * rp2_inv; r3 = (nib & 8u) ? r3 : r3 * rp2_inv;
                                    }
                                } else
                                if (p.res_act) {
                                    r0 = r0 > 0.f ? r0 : r0 * res_inv; r1 = r1 > 0.f ? r1 : r1 * res_inv;
                                    r2 = r2 > 0.f ? r2 : r2 * res_inv; r3 = r3 > 0.f ? r3 : r3 * res_inv;
                                }
                                acc16[i][2 * jh + jj][0] += r0; acc16[i][2 * jh + jj][1] += r1;
                                acc16[i][2 * jh + jj][2] += r2; acc16[i][2 * jh + jj][3] += r3;
                            }
                        }
                    }
                }
            }
        };
        if constexpr (F8S) {
        // ===== block-scaled fp8: a step is ONE tap of a double chunk (128 channels): 16 MFMAs of 32 cycles per wave.  Two patch
        // images (chunks 2c, 2c + 1: 2 x 22.5 KB) filled by LDS-DMA from the pre-quantised copy, a 2-stage ring of per-tap weight
        // stages [2 chunk halves][BN rows][64 B] (2 x 16 KB at BN = 128) -- 77 KB, two blocks per CU; prefetch distance 1, the
        // patch pair re-filled behind an extra barrier at each double-chunk boundary (covered by the other block's waves).
        constexpr int BST8 = 2 * BN * 64;
        constexpr int NJ8 = BST8 / 1024;                 // DMA wave-instructions per stage: 16 / 8
        constexpr int NW8 = NJ8 / 8;                     // per wave: 2 / 1
        unsigned char* Bst8 = smem + 2 * ABY;
        const unsigned char* __restrict__ w8 = (const unsigned char*)p.wgt;
        unsigned d8_off[NW8];
#pragma unroll
        for (int q = 0; q < NW8; ++q) {
            const int j = wid + 8 * q;
            const int pos = j * 64 + lane, row = pos >> 2, slot = pos & 3;
            const int half = (j * 16) / BN, rl = row - half * BN;          // a wave-instruction's 16 rows lie in one half
            const int c = min(nt * BN + rl, p.co - 1);
            d8_off[q] = (unsigned)(half * p.co + c) * 64u + (unsigned)((slot ^ a_swz16(row)) << 4);   // halves = consecutive k-chunks
        }
        const int steps = (kchunks >> 1) * 9;
        auto dma_w8 = [&](int st) __attribute__((always_inline)) {
            const int dc = st / 9, tp = st - dc * 9;
            const unsigned char* slab = w8 + ((long long)p_w(p.tap_wi, tp) * kchunks + 2 * dc) * p.co * 64;
#pragma unroll
            for (int q = 0; q < NW8; ++q) {
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(2 * ABY + (st & 1) * BST8 + (wid + 8 * q) * 1024));
#if !defined(UPS_ABLATE_DMA)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(lds_dst), "v"(d8_off[q]), "s"(slab) : "memory", "m0");
#endif
            }
        };
        auto dma_patch8 = [&](int dc) __attribute__((always_inline)) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned char* base = in8_o + (2 * dc + half) * 64;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (wid + 8 * q >= (PW * PWPS + 15) / 16) continue;        // piece 23 of 22.5 (wave-uniform)
                    const unsigned lds_dst = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)(half * ABY + (wid + 8 * q) * 1024));
#if !defined(UPS_ABLATE_GLOAD)
                    asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                                 :: "s"(pd_mask[q]), "s"(lds_dst), "v"(pd_off[q]), "s"(base) : "memory", "m0");
#endif
                }
            }
        };
        dma_patch8(0);
        dma_w8(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int st = 0; st < steps; ++st) {
            const int n1 = st + 1;
            if (n1 < steps) dma_w8(n1);
            const int tp = st % 9;
            const int dx1 = p_dx(p.tap_off, tp) + 1;
            const int at = arow16[0] + ((p_dy(p.tap_off, tp) + 1) * PWPS + dx1) * APX + a_lane16 + ((q16 ^ a_swz16(px_l16 + dx1)) << 4);
            f8s_tap<TM16, TN16, PWPS * APX, (int)ABY, BN * 64, F8K == 2, (TM16 * TN16 == 16 && TM16 == 4 ? 1 : 2)>(Abuf, Bst8 + (st & 1) * BST8 + (wn * TN * 32) * 64 + boff16, at, acc16);
            if (n1 < steps && n1 % 9 == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // every wave has read the last tap of this patch pair
                dma_patch8(n1 / 9);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        } else {
        UPS_PHASE(1);
        // the tile's sign bytes (PatchK.sgn_res): [256 px][BN / 8], zero outside a ragged image and past the tensor's channels.  REQUESTED
        // first -- they come from HBM, the weights from L2 -- and committed to LDS behind the patch / weight requests, so that their round
        // trip runs under the patch's (two registers for the length of the prologue)
        constexpr int SG_BPP = BN / 8, SG_TPP = SG_BPP >= 8 ? SG_BPP / 8 : 1, SG_PCB = SG_BPP >= 8 ? 8 : 4;     // bytes / threads per pixel, bytes per thread
        const bool sgn_on = SUB == TS && p.sgn_res;
        uint2 sgn_v = make_uint2(0u, 0u);
        const int sgn_px = tid / SG_TPP, sgn_hf = tid - sgn_px * SG_TPP;
        if (sgn_on && sgn_px < 256) {
            const int yy = ty0 + (sgn_px >> 4), xx = tx0 + (sgn_px & 15);
            const int boff = (nt * BN >> 3) + SG_PCB * sgn_hf, rem = (p.ldd >> 3) - boff;       // bytes of the pixel's row from this thread's piece on
            const unsigned char* row = p.dact_bits + (((long long)img_pm * p.h + yy) * p.w + xx) * (p.ldd >> 3) + boff;
            if (yy < p.h && xx < p.w) {
                if (RP2 || (SG_BPP >= 8 && rem >= 8)) sgn_v = *(const uint2*)row;     // (RP2: whole 128-channel N-tiles, launcher)
                else if (rem >= 4) sgn_v.x = *(const unsigned*)row;
            }
        }
        if constexpr (DMAP) dma_patch(0);
        else { load_patch(0); store_patch(Abuf); }
        dma_w(0);
        if (sgn_on && sgn_px < 256) {
            if constexpr (SG_BPP >= 8) *(uint2*)(smem + p.sgn_off + sgn_px * SG_BPP + 8 * sgn_hf) = sgn_v;
            else *(unsigned*)(smem + p.sgn_off + sgn_px * SG_BPP) = sgn_v.x;
        }
        if (OCC != 2 && total > 1) dma_w(1);
        // a single channel chunk on the 3-stage ring: all nine taps' weights fit the ring at once -- everything is requested
        // up front, ONE wait + barrier, then the three tap-rows run back to back (the thin 128x128 layers of encoder_1 on the
        // part images and the heads' input gradients are chains of load -> barrier round trips otherwise)
        const bool one_shot = OCC != 2 && kchunks == 1;
        if (one_shot || CHUNKST) dma_w(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        UPS_PHASE(2);
        if (one_shot) {
            add_res_patch(Abuf, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const unsigned char* B = Bst + g * BST + (wn * TN * 32) * 64 + boff16;
                if constexpr (TAPS != 0) {
                    const int po0 = ((TAPS == 1 ? g : 2 - g) * PWPS + (TAPS == 1 ? 0 : 2)) * APX;
                    bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(Abuf, B, arow16, po0, po0 + (TAPS == 1 ? APX : -APX),
                                            po0 + (TAPS == 1 ? 2 * APX : -2 * APX), TAPS == 1 ? swx0 : swx2, swx1, TAPS == 1 ? swx2 : swx0,
                                            BN * 64, acc16);
                } else {
                    const int dx0 = p_dx(p.tap_off, 3 * g) + 1, dx1 = p_dx(p.tap_off, 3 * g + 1) + 1, dx2 = p_dx(p.tap_off, 3 * g + 2) + 1;
                    const int po0 = ((p_dy(p.tap_off, 3 * g) + 1) * PWPS + dx0) * APX;
                    const int po1 = ((p_dy(p.tap_off, 3 * g + 1) + 1) * PWPS + dx1) * APX;
                    const int po2 = ((p_dy(p.tap_off, 3 * g + 2) + 1) * PWPS + dx2) * APX;
                    bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(Abuf, B, arow16, po0, po1, po2,
                                            a_lane16 + ((q16 ^ a_swz16(px_l16 + dx0)) << 4), a_lane16 + ((q16 ^ a_swz16(px_l16 + dx1)) << 4),
                                            a_lane16 + ((q16 ^ a_swz16(px_l16 + dx2)) << 4), BN * 64, acc16);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // (the epilogue reuses the buffers)
        } else
        if constexpr (OCC == 2) {
        // two blocks per CU: prefetch distance 1 on a 2-stage ring, one patch buffer (re-staged behind an extra barrier at
        // each channel-chunk boundary); the stalls this exposes are covered by the other block's waves
        for (int it = 0; it < total; ++it) {
            const int cc = it / 3, g = it - cc * 3;
            const int n1 = it + 1;
            if (n1 < total) {
                if constexpr (!DMAP) { if (n1 % 3 == 0) load_patch(n1 / 3); }
#if defined(UPS_ABLATE_PATCHWAIT)     // (timing experiment of round 6, results garbage: the next chunk's patch requested a whole chunk ahead INTO THE
                if constexpr (DMAP) { if (g == 0 && cc + 1 < kchunks) dma_patch(cc + 1); }      // BUFFER BEING READ, no re-staging barrier: what a second patch buffer would buy)
#endif
                dma_w(n1);
            }
            const unsigned char* A = Abuf;
            const unsigned char* B = Bst + (it % NST) * BST + (wn * TN * 32) * 64 + boff16;
            if (g == 0) add_res_patch(A, cc);
            if constexpr (TAPS != 0) {
                // static tap geometry: tap-row g reads patch rows g .. (forward) / 2-g .. (flipped), its three taps the column
                // shifts 0, 1, 2 / 2, 1, 0: two scalar operations instead of the tap decode, loop-invariant lane terms
                const int po0 = ((TAPS == 1 ? g : 2 - g) * PWPS + (TAPS == 1 ? 0 : 2)) * APX;
                bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po0 + (TAPS == 1 ? APX : -APX),
                                        po0 + (TAPS == 1 ? 2 * APX : -2 * APX), TAPS == 1 ? swx0 : swx2, swx1, TAPS == 1 ? swx2 : swx0,
                                        BN * 64, acc16);
            } else {
            const int dx0 = p_dx(p.tap_off, 3 * g) + 1, dx1 = p_dx(p.tap_off, 3 * g + 1) + 1, dx2 = p_dx(p.tap_off, 3 * g + 2) + 1;
            const int po0 = ((p_dy(p.tap_off, 3 * g) + 1) * PWPS + dx0) * APX;
            const int po1 = ((p_dy(p.tap_off, 3 * g + 1) + 1) * PWPS + dx1) * APX;
            const int po2 = ((p_dy(p.tap_off, 3 * g + 2) + 1) * PWPS + dx2) * APX;
            bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po1, po2, a_lane16 + ((q16 ^ a_swz16(px_l16 + dx0)) << 4),
                                    a_lane16 + ((q16 ^ a_swz16(px_l16 + dx1)) << 4),
                                    a_lane16 + ((q16 ^ a_swz16(px_l16 + dx2)) << 4), BN * 64, acc16);
            }
#if !defined(UPS_ABLATE_PATCHWAIT)
            if (n1 < total && n1 % 3 == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(UPS_ABLATE_BARRIER)
                __builtin_amdgcn_s_barrier();                 // every wave has read the last tap of this chunk's patch
#endif
                if constexpr (DMAP) dma_patch(n1 / 3);
                else {
#if !defined(UPS_ABLATE_LSTORE)
                store_patch(Abuf);
#endif
                }
            }
#endif
#if !defined(UPS_ABLATE_WWAIT)            // (timing experiment of round 6, results garbage: the wait for the next tap-row's weights)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(UPS_ABLATE_BARRIER)          // (timing experiment of round 6 on THIS loop -- the dominant instances run it; results garbage)
            __builtin_amdgcn_s_barrier();
#endif
        }
        } else if constexpr (CHUNKST) {
        for (int cc = 0; cc < kchunks; ++cc) {
            if (cc + 1 < kchunks) {
                if constexpr (DMAP) dma_patch(cc + 1, ((cc + 1) & 1) * ABY);
                else load_patch(cc + 1);
                dma_w(3 * cc + 3); dma_w(3 * cc + 4); dma_w(3 * cc + 5);
            }
            const unsigned char* A = Abuf + (cc & 1) * ABY;
            add_res_patch(A, cc);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const unsigned char* B = Bst + ((3 * cc + g) % NST) * BST + (wn * TN * 32) * 64 + boff16;
                if constexpr (TAPS != 0) {
                    const int po0 = ((TAPS == 1 ? g : 2 - g) * PWPS + (TAPS == 1 ? 0 : 2)) * APX;
                    bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po0 + (TAPS == 1 ? APX : -APX),
                                            po0 + (TAPS == 1 ? 2 * APX : -2 * APX), TAPS == 1 ? swx0 : swx2, swx1, TAPS == 1 ? swx2 : swx0,
                                            BN * 64, acc16);
                } else {
                    const int dx0 = p_dx(p.tap_off, 3 * g) + 1, dx1 = p_dx(p.tap_off, 3 * g + 1) + 1, dx2 = p_dx(p.tap_off, 3 * g + 2) + 1;
                    const int po0 = ((p_dy(p.tap_off, 3 * g) + 1) * PWPS + dx0) * APX;
                    const int po1 = ((p_dy(p.tap_off, 3 * g + 1) + 1) * PWPS + dx1) * APX;
                    const int po2 = ((p_dy(p.tap_off, 3 * g + 2) + 1) * PWPS + dx2) * APX;
                    bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po1, po2,
                                            a_lane16 + ((q16 ^ a_swz16(px_l16 + dx0)) << 4), a_lane16 + ((q16 ^ a_swz16(px_l16 + dx1)) << 4),
                                            a_lane16 + ((q16 ^ a_swz16(px_l16 + dx2)) << 4), BN * 64, acc16);
                }
            }
            if constexpr (!DMAP) { if (cc + 1 < kchunks) store_patch(Abuf + ((cc + 1) & 1) * ABY); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        } else {
        for (int it = 0; it < total; ++it) {
            const int cc = it / 3, g = it - cc * 3;
            const int n2 = it + 2, n1 = it + 1;
            if (n2 < total) {
                // the patch first (register loads, or DMA pieces straight into the buffer chunk n2/3 will be read from -- free
                // since chunk n2/3 - 2): it stays OLDER than this tap-row's weight DMAs, so the counted wait below covers it
                if (n2 % 3 == 0) {
                    if constexpr (DMAP) dma_patch(n2 / 3, ((n2 / 3) & 1) * ABY);
                    else load_patch(n2 / 3);
                }
                dma_w(n2);
            }
            const unsigned char* A = Abuf + (cc & 1) * ABY;
            const unsigned char* B = Bst + (it % 3) * BST + (wn * TN * 32) * 64 + boff16;
            if (g == 0) add_res_patch(A, cc);
            if constexpr (TAPS != 0) {
                const int po0 = ((TAPS == 1 ? g : 2 - g) * PWPS + (TAPS == 1 ? 0 : 2)) * APX;
                bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po0 + (TAPS == 1 ? APX : -APX),
                                        po0 + (TAPS == 1 ? 2 * APX : -2 * APX), TAPS == 1 ? swx0 : swx2, swx1, TAPS == 1 ? swx2 : swx0,
                                        BN * 64, acc16);
            } else {
            const int dx0 = p_dx(p.tap_off, 3 * g) + 1, dx1 = p_dx(p.tap_off, 3 * g + 1) + 1, dx2 = p_dx(p.tap_off, 3 * g + 2) + 1;
            const int po0 = ((p_dy(p.tap_off, 3 * g) + 1) * PWPS + dx0) * APX;
            const int po1 = ((p_dy(p.tap_off, 3 * g + 1) + 1) * PWPS + dx1) * APX;
            const int po2 = ((p_dy(p.tap_off, 3 * g + 2) + 1) * PWPS + dx2) * APX;
            bf16_taps16<T, TM16, TN16, (SUB == TS ? PWPS * APX : 0), F8, A2FR>(A, B, arow16, po0, po1, po2, a_lane16 + ((q16 ^ a_swz16(px_l16 + dx0)) << 4),
                                    a_lane16 + ((q16 ^ a_swz16(px_l16 + dx1)) << 4),
                                    a_lane16 + ((q16 ^ a_swz16(px_l16 + dx2)) << 4), BN * 64, acc16);
            }
#if !defined(UPS_ABLATE_LSTORE)
            // (the activation patch goes through registers for the fused activation / zero padding; hipcc waits
            // vmcnt(0) for it, which also drains the DMAs once per channel chunk -- measured cost ~0.2 ms of 3.2 ms)
            if constexpr (!DMAP) { if (n1 < total && n1 % 3 == 0) store_patch(Abuf + ((n1 / 3) & 1) * ABY); }
#endif
            // the weights of tap-row it+1 must have landed; the NW DMAs of tap-row it+2 may stay in flight
            if (n2 < total) {
                if (NW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (NW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(UPS_ABLATE_BARRIER)          // (timing experiment of round 6: what the per-tap-row barrier costs; results are garbage without it)
            __builtin_amdgcn_s_barrier();
#endif
        }
        }
        }       // (!F8S)
    } else {
    // ===== fp32 (parity mode): register-staged weights, prefetch distance 2, one barrier per tap-row
    load_patch(0); load_w(ws0, 0, 0);
    store_patch(Abuf); store_w(ws0, Bbuf);
    if (total > 1) load_w(ws1, 1, 0);
    __syncthreads();
    auto iter = [&](int it, WSet& ld_set, const WSet& st_set) __attribute__((always_inline)) {
        const int cc = it / 3, g = it - cc * 3;
        const int n2 = it + 2, n1 = it + 1;
        if (n2 < total) {
            const int c2 = n2 / 3, g2 = n2 - c2 * 3;
            load_w(ld_set, g2, c2);
            if (g2 == 0) load_patch(c2);
        }
        const unsigned char* A = Abuf + (cc & 1) * ABY + a_lane_off;
        const unsigned char* B = Bbuf + (it & 1) * B_BYTES + b_lane_off;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            const int tp = 3 * g + tl;
            const int dxp = p_dx(p.tap_off, tp) + 1;
            const int po = ((p_dy(p.tap_off, tp) + 1) * PWPS + dxp) * APX;
            PMma<float>::template tap<TM, TN, PWPS>(A + po, hh, a_swz(px_l + dxp), B + tl * BN * RS, acc);
        }
        if (n1 < total) {
            store_w(st_set, Bbuf + (n1 & 1) * B_BYTES);
            if (n1 % 3 == 0) store_patch(Abuf + ((n1 / 3) & 1) * ABY);
        }
        __syncthreads();
    };
    for (int it = 0; it < total; it += 2) {
        iter(it, ws0, ws1);
        if (it + 1 < total) iter(it + 1, ws1, ws0);
    }
    }

    UPS_PHASE(3);
    // ---- epilogue
#if defined(UPS_EPI_PRIO)
    __builtin_amdgcn_s_setprio(UPS_EPI_PRIO);      // (round-6 experiment: the epilogue's vector instructions ahead of the CU neighbour's MFMA bursts)
#endif
#if defined(UPS_ABLATE_EPI)
    {   // ablation build: keep the accumulators alive, write (almost) nothing
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += sizeof(T) == 2 ? acc16[2 * i + (e >> 3)][2 * j + ((e >> 2) & 1)][e & 3] : acc[i][j][e];
        if (sacc == 12345.678f) ((float*)p.out)[0] = sacc;
        return;
    }
#endif
    if constexpr (F8) {
        // max |act(x)| of the launch (blocks of the first N-tile; 64 slots spread the atomics): next launch's scale
        if constexpr (!PRE) {
            const float m = wave_max(f8_amax_t);
            if (nt == 0 && lane == 0) ups_amax_slot(p.f8_amax + (bid & 63), m);
        }
        // dequantise: acc = sum (s_a x)(s_w[c] w)  ->  * 1 / (s_a s_w[c])
        const float inv_sa = 1.f / f8_sa;
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = nt * BN + (wn * 2 * TN + j) * 16 + 4 * (lane >> 4) + e;
                const float dq = col < p.co ? p.f8_deq[col] * inv_sa : 0.f;
#pragma unroll
                for (int i = 0; i < 2 * TM; ++i) acc16[i][j][e] *= dq;
            }
    }
    T* __restrict__ outT = (T*)p.out;
    float* __restrict__ outF = (float*)p.out;
    // instances that can write an fp8 copy of their output: all but the 128-wide two-blocks-per-CU ones with static taps / DMA patch
    // (the hot layers' instances, at their register limit); the descriptor-tap form of that shape emits -- it serves the thin
    // single-chunk launches (the logit convolution's input gradient, a store-bound launch whose copy feeds the dominant layer)
    constexpr bool EMITS = __is_same(T, bf16) && (F8 != 0 || !(OCC == 2 && BN == 128) || (TAPS == 0 && !DMAP));
    unsigned char* __restrict__ of8 = p.out_f8;
    float of8_amax = 0.f;
    const float of8_s = (p.out_f8 && p.out_f8_scale) ? *p.out_f8_scale : 1.f;
    const float of8_ns = ups_slope_eff(p.out_f8_act, p.act_slope);
    const T* __restrict__ res = (DMAP && p.res_patch) ? nullptr : (const T*)p.res;
    const T* __restrict__ dact = (const T*)p.dact;
    // the block's first image as uniform (scalar) bases; tile pixel index (ty * 16 + tx) -> 32-bit pixel index from there
    // (the launcher checks that an image group stays below 2^31 bytes in every tensor)
    const long long img_pix = (long long)img_pm * p.h * p.w * (p.d2s ? 4 : 1);
    outT += img_pix * p.ldo; outF += img_pix * p.ldo;
    if (of8) of8 += img_pix * p.ldo;
    if (res) res += img_pix * p.ldr;
    if (dact) dact += img_pix * p.ldd;
    auto gpix = [&](int q) -> unsigned {
        const int ty = q >> 4, tx = q & 15;
        if constexpr (SUB == TS) return (unsigned)((ty0 + ty) * p.w + tx0 + tx);
        else return (unsigned)(((ty / SUB) * G + tx / SUB) * (SUB * SUB) + (ty % SUB) * SUB + (tx % SUB));
    };
    // element offset of GEMM channel chn of tile pixel q in a tensor of ld physical channels (plain or depth-to-space)
    auto gaddr = [&](int q, int chn, int ld) __attribute__((always_inline)) -> unsigned {
        if (SUB == TS && p.d2s) {
            const int cls = chn >> p.d2s_shift, c = chn - (cls << p.d2s_shift);
            const int y2 = 2 * (ty0 + (q >> 4)) + (cls >> 1), x2 = 2 * (tx0 + (q & 15)) + (cls & 1);
            return (unsigned)((y2 * (2 * p.w) + x2) * ld + c);
        }
        return gpix(q) * (unsigned)ld + (unsigned)chn;
    };
    // ragged images (round 5: height / width not a multiple of the 16-pixel tile -- the 56 / 28 / 14-pixel maps of the perceptual
    // trunk behind a 224 x 224 crop): the tiles of the last tile row / column hang over the image.  The staging side needs nothing
    // (patch pixels outside the image are zero, as the halo of every border tile is); the epilogue skips their pixels.
    const bool ragged = SUB == TS && ((p.h & (TS - 1)) | (p.w & (TS - 1))) != 0;
    auto pix_ok = [&](int q) __attribute__((always_inline)) -> bool {
        return !ragged || (ty0 + (q >> 4) < p.h && tx0 + (q & 15) < p.w);
    };
    auto ycoord = [&](int ty) -> int { return SUB == TS ? ty0 + ty : ty % SUB; };
    auto xcoord = [&](int tx) -> int { return SUB == TS ? tx0 + tx : tx % SUB; };

    // bf16 fast path: the residual / activation-derivative tiles come in and the output tile goes out through LDS with
    // 16-byte, fully coalesced accesses (a lane-per-column 2-byte epilogue runs the 1 GB residual read at < 1 TB/s).
    if constexpr (sizeof(T) == 2) {
        if (!p.out_f32 && (p.ldo & 7) == 0 && (p.co_fill & 7) == 0) {
            constexpr int ERS = BN * 2 + 16;                 // staged row stride (bytes)
            constexpr int CPR = BN / 8;                      // 16-byte chunks per row
            constexpr int NIT = 256 * CPR / 512;             // chunks per thread
            unsigned char* R0 = smem;                        // residual tile, then the output tile (in place)
            // activation-derivative tile as sign bits: [256 px][BN / 8 bytes]; res_patch == 2: the copy the prologue left behind the buffers
            const bool sgn_resident = SUB == TS && F8 < 3 && p.sgn_res;
            unsigned char* R1 = sgn_resident ? smem + p.sgn_off : smem + 256 * ERS;
            const int c_lim = p.co_fill - nt * BN;           // valid channels of this N-tile (multiple of 8)
            if (res || (dact && !sgn_resident)) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int idx = tid + 512 * i, px = idx / CPR, ch = idx - px * CPR;
                    if (ch * 8 < c_lim && pix_ok(px)) {
                        if (res) *(uint4*)(R0 + px * ERS + ch * 16) = *(const uint4*)(res + gaddr(px, nt * BN + ch * 8, p.ldr));
                        if (sgn_resident) {
                        } else if (dact && p.dact_bits) {
                            // round 5: the byte the loop below would derive from 16 bytes of the forward input arrives packed (one bit per
                            // element, written by the tensor's producer): 1.07 GB less per launch of the roofline layer, no compares
                            R1[px * CPR + ch] = p.dact_bits[((unsigned long long)img_pix + gpix(px)) * (unsigned)(p.ldd >> 3) +
                                                            (unsigned)((nt * BN + ch * 8) >> 3)];
                        } else if (dact) {
                            const uint4 dv = *(const uint4*)(dact + gaddr(px, nt * BN + ch * 8, p.ldd));
                            R1[px * CPR + ch] = (unsigned char)ups_sign_byte(dv);      // bit e = (element e > 0), elements = the 16-bit halves of the words
                        }
                    }
                }
                __syncthreads();
            }
            UPS_PHASE(4);
            // Accumulator element e of acc16[i][j], lane (p16 = lane & 15, q16 = lane >> 4), is tile pixel (row wm*TM16 + i,
            // column p16), channel (wn*TN16 + j)*16 + 4*q16 + e of the N-tile: 4 consecutive channels per lane, so the
            // residual comes in and the result goes out with one 8-byte LDS access per (i, j).  The CoordConv term of an
            // interior pixel (all nine taps valid) is affine in (x, y): its x part is folded with the bias into one term per
            // channel, the y part is one multiply-add; only tiles that touch the image border look the class table up, and
            // only for their border pixels.
            // (multi-image tiles: every pixel takes the class-table path; the folded terms are then unused)
            const bool all_valid = p.co == p.co_fill;
            const bool has_affine = p.bias != nullptr || p.coord_tab != nullptr;
            const bool epi_vec = (p.co & 3) == 0 && ((((unsigned long long)p.bias) | ((unsigned long long)p.coord_tab)) & 15ull) == 0;
            // (round 6, late: the folded terms are those of the lane's own COLUMN class in an interior row -- class 7 * 8 + xm, 63 for an
            // interior column -- so that only the first and the last ROW of the image take the per-pixel table path below.  Until then every
            // pixel of a first / last column did, in every row of the 28 border tiles of 64: sixteen dependent dword loads per (i, j), and
            // the forward's epilogue took 11.2 us on average, 21 us at p90, against the input gradient's 7.3)
            const bool border_tile = p.coord_tab && (SUB < TS || ty0 == 0 || ty0 + TS >= p.h);
            const float xf = (float)(tx0 + p16);
            const int xq = xcoord(p16);
            const int xm = (xq > 0 ? 1 : 0) | 2 | (xq + 1 < p.w ? 4 : 0);
            const int cls_col = SUB == TS ? 7 * 8 + xm : 63;          // (multi-image tiles: every pixel takes the table path anyway)
#pragma unroll
            for (int j = 0; j < TN16; ++j) {
                const int cl = (wn * TN16 + j) * 16 + 4 * q16;
                const int col = nt * BN + cl;
                if (col >= p.co_fill) continue;          // co_fill is a multiple of 8: the 4 channels go together
                float xs[4], t2v[4];
                bool cv[4];
                if (epi_vec) {
                    // the lane's four channels are consecutive and 16-byte aligned in the bias vector and in every row of the class
                    // table: one 16-byte load each instead of four (the 128-wide instances issued 64 dword loads per lane here)
                    const bool ok = col < p.co;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), t04 = b4, t14 = b4, t24 = b4;
                    if (ok && p.bias) b4 = *(const float4*)(p.bias + col);
                    if (ok && p.coord_tab) {
                        const float* tb = p.coord_tab + (long long)cls_col * 3 * p.co + col;
                        t04 = *(const float4*)tb; t14 = *(const float4*)(tb + p.co); t24 = *(const float4*)(tb + 2 * p.co);
                    }
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, a0[4] = {t04.x, t04.y, t04.z, t04.w};
                    const float a1[4] = {t14.x, t14.y, t14.z, t14.w}, a2[4] = {t24.x, t24.y, t24.z, t24.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cv[e] = ok; t2v[e] = a2[e]; xs[e] = fmaf(xf, a1[e], bb[e] + a0[e]); }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cv[e] = col + e < p.co;
                    const float bias = (cv[e] && p.bias) ? p.bias[col + e] : 0.f;
                    float t0 = 0.f, t1 = 0.f;
                    t2v[e] = 0.f;
                    if (cv[e] && p.coord_tab) {
                        const float* tb = p.coord_tab + (long long)cls_col * 3 * p.co + col + e;
                        t0 = tb[0]; t1 = tb[p.co]; t2v[e] = tb[2 * p.co];
                    }
                    xs[e] = fmaf(xf, t1, bias + t0);
                }
                }
                const int dsh = cl & 4;                  // the lane's 4 sign bits inside the tile's derivative byte
#pragma unroll
                for (int i = 0; i < TM16; ++i) {
                    const int yrow = wm * TM16 + i;
                    const int px = yrow * 16 + p16;
                    float v[4];
                    if (has_affine) {          // (uniform: an input gradient has neither bias nor CoordConv term)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc16[i][j][e] + xs[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc16[i][j][e];
                    }
                    if (p.coord_tab) {
                        const float yf = (float)(ty0 + yrow);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += yf * t2v[e];
                        if (border_tile) {
                            const int y = ycoord(yrow);
                            const int ym = (y > 0 ? 1 : 0) | 2 | (y + 1 < p.h ? 4 : 0);
                            if (ym != 7 || SUB < TS) {
                                if (epi_vec && cv[0]) {      // (the lane's four channels with one 16-byte load per term: every pixel of the
                                                             // multi-image tiles -- the 8 x 8 / 4 x 4 maps -- comes through here)
                                    const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col;
                                    const float4 t04 = *(const float4*)tb, t14 = *(const float4*)(tb + p.co), t24 = *(const float4*)(tb + 2 * p.co);
                                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                                    if (p.bias) b4 = *(const float4*)(p.bias + col);
                                    const float a0[4] = {t04.x, t04.y, t04.z, t04.w}, a1[4] = {t14.x, t14.y, t14.z, t14.w};
                                    const float a2[4] = {t24.x, t24.y, t24.z, t24.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = acc16[i][j][e] + bb[e] + (a0[e] + (float)xq * a1[e] + (float)y * a2[e]);
                                } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (!cv[e]) continue;
                                    const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col + e;
                                    const float bias = p.bias ? p.bias[col + e] : 0.f;
                                    v[e] = acc16[i][j][e] + bias + (tb[0] + (float)xq * tb[p.co] + (float)y * tb[2 * p.co]);
                                }
                                }
                            }
                        }
                    }
                    if (dact) {
                        const unsigned db = (unsigned)R1[px * CPR + (cl >> 3)] >> dsh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= (db >> e) & 1u ? 1.f : dact_ns;
                    }
                    uint2* slot = (uint2*)(R0 + px * ERS + cl * 2);
                    if (res) {
                        const uint2 rv = *slot;
                        float r0, r1, r2, r3;
                        ups_unpack2<T>(rv.x, r0, r1); ups_unpack2<T>(rv.y, r2, r3);
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
                    if (!all_valid) {          // (uniform: pad channels exist only when co < co_fill)
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!cv[e]) v[e] = 0.f;
                    }
                    *slot = make_uint2(Chunk<T>::pk(v[0], v[1]), Chunk<T>::pk(v[2], v[3]));
                }
            }
            __syncthreads();
            UPS_PHASE(5);
            if (p.mask_grad) {
                // input gradient of the part-masked convolution, reduced to the hard mask: g_hard[b][y][x][part] =
                // sum_c gx[c] * view[b][y][x][c] with gx rounded to the activation dtype first (as the tensor it replaces was)
                if (SUB == TS && tid < 256 && __is_same(T, bf16) && pix_ok(tid)) {
                    const long long pixb = (long long)img * p.h * p.w + (long long)(ty0 + (tid >> 4)) * p.w + tx0 + (tid & 15);
                    const bf16* gv = (const bf16*)(R0 + tid * ERS);
                    const float* vv = p.mask_view + pixb * p.co;
                    float sacc = 0.f;
                    for (int c = 0; c < p.co; ++c) sacc += (float)gv[c] * vv[c];
                    p.mask_grad[pixb * p.mask_P + part] = sacc;
                }
                return;
            }
            // (round 6: the epilogue's vector instructions run on the issue port the CU neighbour's MFMAs need, DESIGN section 3.  A thread's
            // NIT pieces are 512 / CPR pixels = whole tile rows apart: on plain full tiles their addresses are one base + a constant step
            // instead of a 64-bit multiply-add chain per piece)
            const bool lin = SUB == TS && !p.d2s && !ragged;
            const unsigned px0 = (unsigned)tid / CPR, ch0 = (unsigned)tid - px0 * CPR;
            const unsigned ga0 = lin ? gaddr((int)px0, nt * BN + (int)ch0 * 8, p.ldo) : 0u;
            const unsigned ga_step = (unsigned)((512 / CPR / 16) * p.w) * (unsigned)p.ldo;       // 512 / CPR pixels = (512 / CPR) / 16 tile rows
            const unsigned long long sg0 = lin && p.sign_out ? ((unsigned long long)img_pix + gpix((int)px0)) * (unsigned)(p.ldo >> 3) +
                                                                (unsigned)((nt * BN + (int)ch0 * 8) >> 3) : 0ull;
            const unsigned sg_step = (unsigned)((512 / CPR / 16) * p.w) * (unsigned)(p.ldo >> 3);
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int idx = tid + 512 * i, px = idx / CPR, ch = idx - px * CPR;
                if (ch * 8 < c_lim && pix_ok(px)) {
                    const uint4 u = *(const uint4*)(R0 + px * ERS + ch * 16);
                    const unsigned ga = lin ? ga0 + (unsigned)i * ga_step : gaddr(px, nt * BN + ch * 8, p.ldo);
                    *(uint4*)(outT + ga) = u;
                    if (p.sign_out) {          // (uniform) bit e = stored element e > 0: positive and non-zero as a 16-bit integer
                        const unsigned sb = ups_sign_byte(u);
                        const unsigned long long sgi = lin ? sg0 + (unsigned long long)((unsigned)i * sg_step)
                                                           : ((unsigned long long)img_pix + gpix(px)) * (unsigned)(p.ldo >> 3) + (unsigned)((nt * BN + ch * 8) >> 3);
                        p.sign_out[sgi] = (unsigned char)sb;
                    }
                    // fp8 copy for the consumer (uniform branch; not compiled into the 128-wide bf16 kernel at two blocks per CU, which
                    // has no register to spare -- the launcher keeps producers off it): act(out) -> max -> * scale -> 8 bytes
                    if (EMITS && p.out_f8_amax) {
                        const unsigned wsrc[4] = {u.x, u.y, u.z, u.w};
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f[2 * e] = __uint_as_float(wsrc[e] << 16); f[2 * e + 1] = __uint_as_float(wsrc[e] & 0xffff0000u);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (p.out_f8_act != UPS_ACT_NONE) f[e] = ups_act_ns(f[e], of8_ns);
                            of8_amax = fmaxf(of8_amax, fabsf(f[e]));
                        }
                        if (p.out_f8) {
                            int d0 = 0, d1 = 0;
                            if (p.out_f8_e5m2) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e] * of8_s, -57344.f, 57344.f);
                                d0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], d0, false); d0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], d0, true);
                                d1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], d1, false); d1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], d1, true);
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e] * of8_s, -448.f, 448.f);
                                d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], d0, false); d0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], d0, true);
                                d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], d1, false); d1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], d1, true);
                            }
                            *(uint2*)(of8 + ga) = make_uint2((unsigned)d0, (unsigned)d1);
                        }
                    }
                }
            }
            if (EMITS && p.out_f8_amax) {
                const float m = wave_max(of8_amax);
                if (lane == 0) ups_amax_slot(p.out_f8_amax + (bid & 63), m);
            }
            UPS_PHASE(6);
            return;
        }
    }

    // per-element epilogue (fp32 mode, fp32 outputs, unaligned channel counts).  fp32 path: 32x32 accumulator blocks, element e
    // of lane l = pixel rr(e, l >> 5) of the block, channel l & 31; bf16 path: 16x16 blocks, element e = channel 4*q16 + e of
    // pixel column p16 (see above)
    auto emit = [&](float a, int tyq, int txq, int col, float bias) __attribute__((always_inline)) {
        const bool cvalid = col < p.co;
        const int y = ycoord(tyq), x = xcoord(txq);
        if (!pix_ok(tyq * 16 + txq)) return;
        const unsigned pix = gpix(tyq * 16 + txq);
        float v = 0.f;
        if (cvalid) {
            v = a + bias;
            if (p.coord_tab) {
                const int ym = (y > 0 ? 1 : 0) | 2 | (y + 1 < p.h ? 4 : 0);
                const int xm = (x > 0 ? 1 : 0) | 2 | (x + 1 < p.w ? 4 : 0);
                const float* tb = p.coord_tab + (long long)(ym * 8 + xm) * 3 * p.co + col;
                v += tb[0] + (float)x * tb[p.co] + (float)y * tb[2 * p.co];
            }
            if (dact) v *= (ld_as_float<T>(dact + pix * p.ldd + col) > 0.f) ? 1.f : dact_ns;
            if (res) {
                float r = ld_as_float<T>(res + pix * p.ldr + col);
                if (p.res_act) r = r > 0.f ? r : r * res_inv;
                v += r;
            }
            if (p.out_act) v = ups_vmax(v, oact_ns * v);
        }
        if (p.out_f32) outF[pix * p.ldo + col] = v;
        else st_from_float<T>(outT + pix * p.ldo + col, v);
    };
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int tn = 0; tn < TN16; ++tn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = nt * BN + (wn * TN16 + tn) * 16 + 4 * q16 + e;
                if (col >= p.co_fill) continue;
                const float bias = (col < p.co && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
                for (int tm = 0; tm < TM16; ++tm) emit(acc16[tm][tn][e], wm * TM16 + tm, p16, col, bias);
            }
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = nt * BN + (wn * TN + tn) * 32 + (lane & 31);
            if (col >= p.co_fill) continue;
            const float bias = (col < p.co && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    emit(acc[tm][tn][e], (wm * TM + tm) * 2 + (rr >> 4), rr & 15, col, bias);
                }
            }
        }
    }
}

template <typename T, int BN, int OCC, int SUB, int F8 = 0, bool PRE = false, int TAPS = 0, bool DMAP = false, bool CSTD = false>
int launch_bn(const PatchK& k, hipStream_t s) {
    constexpr int EPC = Chunk<T>::N;
    constexpr int G = TS / SUB, PR = G * (SUB + 2), PWPS = (PR + 3) / 4 * 4;
    constexpr size_t ABY = (size_t)PR * PWPS * APX;
    // an fp8 copy of the output can only be written by the instances that compile the emitting store loop (EMITS in the kernel):
    // a launch that asks for one must never land on another instance and leave the copy unwritten
    constexpr bool emits = __is_same(T, bf16) && (F8 != 0 || !(OCC == 2 && BN == 128) || (TAPS == 0 && !DMAP));
    if ((k.out_f8 || k.out_f8_amax) && !emits) return UPS_E_UNSUPPORTED;
    const int tiles_x = SUB == TS ? (k.w + TS - 1) / TS : 1, tiles_y = SUB == TS ? (k.h + TS - 1) / TS : 1;
    const int ntn = ups_cdiv(k.co_fill, BN);
    const int kchunks = ups_cdiv(k.ci, F8 ? 64 : 4 * EPC);
    const int nblocks = (k.n / (G * G)) * tiles_x * tiles_y * ntn;
    PatchK kk = k;
    kk.m_ntn = div_magic(nblocks, ntn); kk.m_parts = div_magic(nblocks, k.mask_P);
    kk.m_tx = div_magic(nblocks, tiles_x); kk.m_ty = div_magic(nblocks, tiles_y);
    constexpr size_t BST = 3 * (size_t)BN * 64;
    const int nabuf = (sizeof(T) == 2 && SUB == TS && (kchunks == 1 || OCC == 2)) ? 1 : 2;
    constexpr int nst = OCC == 2 ? 2 : (((SUB < TS || (CSTD && DMAP && SUB == TS)) && BN <= 64 && F8 == 0 && OCC == 1) ? 6 : 3);      // (NST / CHUNKST of the kernel)
    size_t shmem = sizeof(T) == 2 ? nabuf * ABY + nst * BST : 2 * ABY + 2 * 3 * BN * RS;
    size_t shmem_max = sizeof(T) == 2 ? (OCC == 2 ? 1 : 2) * ABY + nst * BST : shmem;
    if (F8 >= 3) shmem = shmem_max = 2 * ABY + 2 * (2 * (size_t)BN * 64);      // block-scaled fp8: two patch images, two per-tap stages
    const size_t epi = sizeof(T) == 2 ? 256 * (size_t)(BN * 2 + 16) + 256 * (size_t)(BN / 8) : 0;   // staged bf16 epilogue
    if (epi > shmem) shmem = epi;
    constexpr bool rp2_ok = __is_same(T, bf16) && BN == 128 && TAPS == 2 && DMAP && SUB == TS && F8 == 0;
    constexpr size_t sgn_bytes = (sizeof(T) == 2 && SUB == TS && F8 < 3) ? 256 * (size_t)(BN / 8) : 0;       // the resident sign tile (PatchK.sgn_res)
    kk.sgn_res = (sgn_bytes && kk.dact && kk.dact_bits && kk.ldd % 64 == 0 && (((uintptr_t)kk.dact_bits) & 7) == 0 && !kk.d2s) ? 1 : 0;
    if (rp2_ok && kk.co_fill % BN != 0) kk.sgn_res = 0;                    // (that instance's loader reads the 16 bytes of whole N-tiles)
    if (kk.res_patch == 2 && !(rp2_ok && kk.sgn_res)) kk.res_patch = 0;    // (this instance has no such path: the residual in the epilogue)
    kk.sgn_off = (int)shmem;
    if (kk.sgn_res) shmem += sgn_bytes;
    static UpsPerDevice attr_set;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<T, BN, OCC, SUB, F8, PRE, TAPS, DMAP, CSTD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((epi > shmem_max ? epi : shmem_max) + sgn_bytes));
        if (e != hipSuccess) return UPS_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_patch_kernel<T, BN, OCC, SUB, F8, PRE, TAPS, DMAP, CSTD>), dim3(nblocks), dim3(512), shmem, s, kk, tiles_x, tiles_y, ntn,
                       kchunks, nblocks);
    return UPS_OK;
}

static int static_taps_on() {   // UPS_PATCH_STATIC=0: the descriptor-driven tap loop everywhere (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("UPS_PATCH_STATIC"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

static int dma_patch_on() {   // UPS_PATCH_DMA=0: register-staged patch everywhere (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("UPS_PATCH_DMA"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

static int patch_occ() {   // UPS_PATCH_OCC=1 forces the one-block-per-CU configuration (A/B runs)
    static int v = -1;
    if (v < 0) { const char* e = getenv("UPS_PATCH_OCC"); v = (e && e[0] == '1') ? 1 : 2; }
    return v;
}

template <typename T, int SUB>
int launch_small(const PatchK& k, hipStream_t s) {
    // whole 8x8 / 4x4 images packed 4 / 16 to a tile: narrower N-tiles when the grid would not fill the chip
    const int tiles = k.n / ((TS / SUB) * (TS / SUB));
    if (k.co_fill > 64 && tiles * ups_cdiv(k.co_fill, 128) >= 192) return launch_bn<T, 128, 1, SUB>(k, s);
    if (k.co_fill > 32 && tiles * ups_cdiv(k.co_fill, 64) >= 128) return launch_bn<T, 64, 1, SUB>(k, s);
    return launch_bn<T, 32, 1, SUB>(k, s);
}

// one tile per image, bf16 / fp16, no fp8: the variant with the static tap geometry and, where the input needs nothing done to it
// on the way (no activation-on-load, no part mask, whole 32-channel chunks), the halo patch by LDS-DMA
template <typename T, int BN, int OCC>
int launch_v(const PatchK& k, hipStream_t s) {
    const bool dmap = dma_patch_on() && k.act_in == UPS_ACT_NONE && !k.mask && !k.mask_grad && k.ci % 32 == 0;
    if (static_taps_on() && dmap) {
        if (k.taps_static == 1) return launch_bn<T, BN, OCC, TS, 0, false, 1, true>(k, s);
        if (k.taps_static == 2) return launch_bn<T, BN, OCC, TS, 0, false, 2, true>(k, s);
    }
    // (without the DMA patch only the forward order of the widest two-blocks-per-CU instance has a static form: the others spill)
    if (static_taps_on() && k.taps_static == 1 && BN == 128 && OCC == 2) return launch_bn<T, BN, OCC, TS, 0, false, (BN == 128 && OCC == 2) ? 1 : 0>(k, s);
    return launch_bn<T, BN, OCC, TS>(k, s);
}

template <typename T>
int launch_t(const PatchK& k, hipStream_t s) {
    if (k.h == 8) return launch_small<T, 8>(k, s);
    if (k.h == 4) return launch_small<T, 4>(k, s);
    if constexpr (sizeof(T) == 2) {
      if constexpr (__is_same(T, bf16)) {
        if (k.f8_deq && k.in_f8) {      // pre-quantised input: bf16-sized staging, two blocks per CU on large grids
            const int tiles8 = k.n * ((k.w + TS - 1) / TS) * ((k.h + TS - 1) / TS);
            const bool big128 = patch_occ() == 2 && tiles8 * ups_cdiv(k.co_fill, 128) >= 512;
            const bool big64 = patch_occ() == 2 && tiles8 * ups_cdiv(k.co_fill, 64) >= 512;
            // whole 128-channel double chunks on a grid of two blocks per CU: the block-scaled K = 128 MFMA (UPS_F8_SCALED=0: off)
            static int scaled = -1;
            if (scaled < 0) { const char* e = getenv("UPS_F8_SCALED"); scaled = (e && e[0] == '0') ? 0 : 1; }
            if (scaled && k.ci % 128 == 0) {
                if (k.co_fill > 64 && big128) return k.f8_e5m2 ? launch_bn<T, 128, 2, TS, 4, true>(k, s) : launch_bn<T, 128, 2, TS, 3, true>(k, s);
                if (k.co_fill <= 64 && big64) return k.f8_e5m2 ? launch_bn<T, 64, 2, TS, 4, true>(k, s) : launch_bn<T, 64, 2, TS, 3, true>(k, s);
            }
            if (k.f8_e5m2) {
                if (k.co_fill > 64) return big128 ? launch_bn<T, 128, 2, TS, 2, true>(k, s) : launch_bn<T, 128, 1, TS, 2, true>(k, s);
                return big64 ? launch_bn<T, 64, 2, TS, 2, true>(k, s) : launch_bn<T, 64, 1, TS, 2, true>(k, s);
            }
            if (k.co_fill > 64) return big128 ? launch_bn<T, 128, 2, TS, 1, true>(k, s) : launch_bn<T, 128, 1, TS, 1, true>(k, s);
            return big64 ? launch_bn<T, 64, 2, TS, 1, true>(k, s) : launch_bn<T, 64, 1, TS, 1, true>(k, s);
        }
        if (k.f8_deq) {      // fp8 operands (eligibility checked by the caller): e4m3 activations or e5m2 gradients
            if (k.f8_e5m2) {
                if (k.co_fill > 64) return launch_bn<T, 128, 1, TS, 2>(k, s);
                if (k.co_fill > 32) return launch_bn<T, 64, 1, TS, 2>(k, s);
                return launch_bn<T, 32, 1, TS, 2>(k, s);
            }
            if (k.co_fill > 64) return launch_bn<T, 128, 1, TS, 1>(k, s);
            if (k.co_fill > 32) return launch_bn<T, 64, 1, TS, 1>(k, s);
            return launch_bn<T, 32, 1, TS, 1>(k, s);
        }
      }
        // two blocks per CU once the grid has at least two blocks for every CU (smaller grids spread over the chip instead);
        // single-chunk layers (ci <= 32, e.g. the dgrad of the P-channel logit conv) use 64-wide tiles and one patch buffer
        const int tiles = k.n * ((k.w + TS - 1) / TS) * ((k.h + TS - 1) / TS);
        if (k.co_fill > 64 && k.ci > 32) {
            if (patch_occ() == 2 && !k.out_f8_amax && tiles * ups_cdiv(k.co_fill, 128) >= 512) return launch_v<T, 128, 2>(k, s);
            // a grid of one 128-wide block per CU: 64-wide tiles put two blocks on every CU instead (4 waves per SIMD)
            static int mid = -1;
            if (mid < 0) { const char* e = getenv("UPS_PATCH_MID"); mid = (e && e[0] == '0') ? 0 : 1; }
            if (mid && patch_occ() == 2 && tiles * ups_cdiv(k.co_fill, 64) >= 512) return launch_v<T, 64, 2>(k, s);
            {   // at most one 128-wide block per CU, many channel chunks: 64-wide tiles with chunk-granular weight stages (CSTD)
                static int cst = -1;
                if (cst < 0) { const char* e = getenv("UPS_PATCH_CST"); cst = (e && e[0] == '0') ? 0 : 1; }
                const bool dmap = dma_patch_on() && k.act_in == UPS_ACT_NONE && !k.mask && !k.mask_grad && k.ci % 32 == 0;
                if (cst && static_taps_on() && dmap && !k.out_f8_amax && k.ci >= 128 && tiles * ups_cdiv(k.co_fill, 128) <= 256) {
                    if (k.taps_static == 1) return launch_bn<T, 64, 1, TS, 0, false, 1, true, true>(k, s);
                    if (k.taps_static == 2) return launch_bn<T, 64, 1, TS, 0, false, 2, true, true>(k, s);
                }
            }
            return launch_v<T, 128, 1>(k, s);
        }
        if (k.co_fill > 32) {
            // a single input chunk with many outputs (the input gradient of the P-channel logit convolution: 16 -> 256 channels,
            // a store-bound launch): 128-wide tiles halve the number of blocks and patch loads (0.55 -> 0.48 ms; UPS_PATCH_THIN128=0: off)
            static int thin128 = -1;
            if (thin128 < 0) { const char* e = getenv("UPS_PATCH_THIN128"); thin128 = (e && e[0] == '0') ? 0 : 1; }
            // (not when the launch has to write an fp8 copy of its output: the 128-wide two-blocks-per-CU instance cannot -- EMITS)
            if (thin128 && k.co_fill > 64 && patch_occ() == 2 && tiles * ups_cdiv(k.co_fill, 128) >= 512) {
                // a launch that writes an fp8 copy takes the descriptor-tap form of the 128-wide instance (the one that emits)
                if (k.out_f8_amax) return launch_bn<T, 128, 2, TS>(k, s);
                return launch_v<T, 128, 2>(k, s);
            }
            return (patch_occ() == 2 && k.ci > 32 && tiles * ups_cdiv(k.co_fill, 64) >= 512) ? launch_v<T, 64, 2>(k, s)
                                                                                              : launch_v<T, 64, 1>(k, s);
        }
        {   // few outputs over many input chunks (the P-channel logit convolution: 256 -> 10 at 128x128): the 32-wide instance keeps one
            // block per CU busy with one exposed patch round trip per chunk; UPS_PATCH_THINOUT=64 / 128 tries the two-blocks-per-CU
            // instances on it (three quarters / seven eighths of their MFMA columns idle, but the launch is latency-bound)
            static int thinout = -1;
            if (thinout < 0) { const char* e = getenv("UPS_PATCH_THINOUT"); thinout = e ? atoi(e) : 0; }
            if (thinout == 64 && k.ci > 32 && patch_occ() == 2 && tiles >= 512) return launch_v<T, 64, 2>(k, s);
            if (thinout == 128 && k.ci > 32 && patch_occ() == 2 && tiles >= 512 && !k.out_f8_amax) return launch_v<T, 128, 2>(k, s);
        }
        return launch_v<T, 32, 1>(k, s);
    } else {
        if (k.co_fill > 64) return launch_bn<T, 128, 1, TS>(k, s);
        if (k.co_fill > 32) return launch_bn<T, 64, 1, TS>(k, s);
        return launch_bn<T, 32, 1, TS>(k, s);
    }
}

}  // namespace

// Does a launch of this kernel take the staged 16-bit epilogue on a plain lattice -- the one that writes ups_conv_desc.sign_out and
// reads ups_conv_desc.dact_bits?  (conv_igemm.hip packs the signs in a separate pass when the launch that ran did not.)
bool ups_conv3x3_patch_signs(const ups_conv_desc* d) {
    return d->dtype != UPS_F32 && !d->out_f32 && (d->ldo & 7) == 0 && (d->co_fill & 7) == 0 && !d->d2s && !d->mask_grad && !d->mask_bits;
}

// Internal entry used by ups_conv_igemm's dispatcher (conv_igemm.hip). Returns 1 if the problem is not eligible.
int ups_conv3x3_patch_try(const ups_conv_desc* d, hipStream_t s) {
    if (d->ntaps != 9 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->out_oy || d->out_ox)
        return 1;
    if (d->hi != d->ho || d->wi != d->wo) return 1;
    if (d->d2s ? (d->out_h != 2 * d->ho || d->out_w != 2 * d->wo) : (d->out_h != d->ho || d->out_w != d->wo)) return 1;
    // 16-bit image-pitch pixel index of a staged item, 24-bit row pitch for its v_mad_u32_u24
    if ((long long)(TS + 2) * d->wi >= 0xffff || (long long)d->ldi * 4 >= (1 << 24)) return 1;
    {   // 32-bit offsets inside the image group of a block (up to 16 images per tile for the 4x4 case)
        long long ldmax = d->ldi > d->ldo ? d->ldi : d->ldo;
        if (d->res && d->ldr > ldmax) ldmax = d->ldr;
        if (d->dact && d->ldd > ldmax) ldmax = d->ldd;
        if (16ll * d->hi * d->wi * ldmax * 4 >= (1ll << 31)) return 1;
    }
    const bool small = d->hi == d->wi && (d->hi == 8 || d->hi == 4) && d->n % ((TS / d->hi) * (TS / d->hi)) == 0;
    {
        static int small_on = -1;
        if (small_on < 0) { const char* e = getenv("UPS_NO_SMALL_PATCH"); small_on = (e && e[0] == '1') ? 0 : 1; }
        // ragged tiles (UPS_PATCH_RAGGED=0: off): plain forward / input-gradient launches only -- no depth-to-space output, part
        // masks, fp8 operands or copies (their store loops and scale maxima assume whole tiles), at least one whole tile row's worth
        // of pixels so that the launch is not mostly padding
        static int ragged_on = -1;
        if (ragged_on < 0) { const char* e = getenv("UPS_PATCH_RAGGED"); ragged_on = (e && e[0] == '0') ? 0 : 1; }
        const bool ragged_ok = ragged_on && d->hi >= 12 && d->wi >= 12 && !d->d2s && !d->mask_bits && !d->mask_grad && !d->f8_deq &&
                               !d->in_f8 && !d->out_f8 && !d->out_f8_amax;
        if ((d->hi % TS || d->wi % TS) && !(small && small_on) && !ragged_ok) return 1;
    }
    bool seen[9] = {false, false, false, false, false, false, false, false, false};
    for (int t = 0; t < 9; ++t) {
        const int dy = d->tap_dy[t], dx = d->tap_dx[t];
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || d->tap_w[t] < 0 || d->tap_w[t] > 15) return 1;
        seen[(dy + 1) * 3 + dx + 1] = true;
    }
    for (int t = 0; t < 9; ++t) if (!seen[t]) return 1;
    if (d->coord_tab) {   // the epilogue derives the class from the forward tap order r-major, dy = r-1, dx = s-1
        for (int t = 0; t < 9; ++t) if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) return 1;
    }
    PatchK k;
    k.mask = d->mask_bits; k.mask_grad = d->mask_grad; k.mask_view = d->mask_view;
    k.f8_deq = d->f8_deq; k.f8_scale = d->f8_scale; k.f8_amax = d->f8_amax; k.f8_e5m2 = d->f8_e5m2;
    k.in_f8 = (const unsigned char*)d->in_f8; k.out_f8 = (unsigned char*)d->out_f8; k.out_f8_scale = d->out_f8_scale;
    k.out_f8_amax = d->out_f8_amax; k.out_f8_act = d->out_f8_act; k.out_f8_e5m2 = d->out_f8_e5m2;
    if (d->in_f8 && (!d->f8_deq || d->co_fill <= 32)) return 1;
    if ((d->out_f8 || d->out_f8_amax) && (d->dtype != UPS_BF16 || small || d->out_f32 || (d->ldo & 7) || (d->co_fill & 7) || d->mask_grad ||
                                          !d->out_f8_amax || (d->out_f8 && !d->out_f8_scale)))
        return 1;
    k.d2s = 0; k.d2s_shift = 0;
    if (d->d2s) {
        // depth-to-space output: bf16 staged epilogue on 16-aligned lattices, 4 classes of d2s (power of two, >= 8) channels
        if (d->dtype != UPS_BF16 || small || (d->hi % TS) || (d->wi % TS) || d->d2s < 8 || (d->d2s & (d->d2s - 1)) ||
            d->co != 4 * d->d2s || d->co_fill != d->co || d->out_f32 || (d->ldo & 7) || d->mask_bits || d->mask_grad ||
            d->coord_tab || d->bias || d->f8_deq)
            return 1;
        k.d2s = 1;
        while ((1 << k.d2s_shift) < d->d2s) ++k.d2s_shift;
    }
    if (d->f8_deq) {
        if (d->dtype != UPS_BF16 || small || (d->hi % TS) || (d->wi % TS) || d->ci % 64 || d->mask_bits || d->mask_grad ||
            !d->f8_scale || (!d->f8_amax && !d->in_f8))
            return 1;
    }
    k.mask_B = 0; k.mask_P = 1;
    if (d->mask_bits || d->mask_grad) {
        // part mode: bf16, one image per tile, staged epilogue
        if (d->dtype != UPS_BF16 || small || d->mask_batch <= 0 || d->n % d->mask_batch) return 1;
        if (d->mask_bits && d->mask_grad) return 1;
        k.mask_B = d->mask_batch; k.mask_P = d->n / d->mask_batch;
        if (k.mask_P > 32) return 1;
        if (d->mask_grad && (!d->mask_view || d->out_f32 || (d->ldo & 7) || (d->co_fill & 7) || d->co_fill > 32 || d->res || d->dact)) return 1;
    }
    k.n = d->n; k.h = d->hi; k.w = d->wi; k.ci = d->ci; k.ldi = d->ldi; k.co = d->co; k.co_fill = d->co_fill;
    k.ldo = d->ldo; k.ldr = d->ldr; k.ldd = d->ldd; k.act_in = d->act_in; k.out_f32 = d->out_f32;
    k.dact_kind = d->dact_kind; k.has_ctab = d->coord_tab != nullptr; k.act_slope = d->act_slope;
    k.in = d->in; k.wgt = d->w; k.out = d->out; k.bias = d->bias; k.coord_tab = d->coord_tab; k.res = d->res; k.dact = d->dact;
    // sign bits: only the staged 16-bit epilogue writes / reads them (ups_conv3x3_patch_signs says when that one runs)
    k.sign_out = ups_conv3x3_patch_signs(d) ? (unsigned char*)d->sign_out : nullptr;
    k.dact_bits = (d->dact && ups_conv3x3_patch_signs(d) && (d->ldd & 7) == 0) ? (const unsigned char*)d->dact_bits : nullptr;
    bool fwd = true, flip = true;
    {
        for (int t = 0; t < 9; ++t) {
            if (d->tap_w[t] != t) fwd = flip = false;
            if (d->tap_dy[t] != t / 3 - 1 || d->tap_dx[t] != t % 3 - 1) fwd = false;
            if (d->tap_dy[t] != 1 - t / 3 || d->tap_dx[t] != 1 - t % 3) flip = false;
        }
        k.taps_static = fwd ? 1 : (flip ? 2 : 0);
    }
    k.out_act = d->out_act; k.res_act = d->res_act;
    {   // residual block: res == in, channel chunk c <-> output channels 32c.. (no CoordConv channels are part of `in`), whole chunks,
        // 16-bit, one tile per image, an N-tile of at least two 16-channel blocks per wave (BN >= 64 in launch_t's choice below)
        static int rp_on = -1;
        if (rp_on < 0) { const char* e = getenv("UPS_RES_PATCH"); rp_on = (e && e[0] == '0') ? 0 : 1; }
        k.res_patch = rp_on && d->res && d->res == d->in && d->ldr == d->ldi && d->ci == d->co_fill && d->co == d->co_fill && d->ci % 32 == 0 &&
                      d->dtype != UPS_F32 && !small && !d->dact && !d->mask_bits && !d->mask_grad && !d->d2s && !d->f8_deq && !d->out_f32 &&
                      d->act_in == UPS_ACT_NONE &&    // (with activation-on-load the patch holds act(x), the residual wants x)
                      dma_patch_on() && static_taps_on() && (fwd || flip);       // ... and only the DMA-patch instances implement it
        // round 6: the input gradient of a residual block whose act' arrives as sign bytes (leaky ReLU: 1 / slope exists)
        static int rp2_on = -1;
        if (rp2_on < 0) { const char* e = getenv("UPS_RES_PATCH_DGRAD"); rp2_on = (e && e[0] == '0') ? 0 : 1; }
        if (rp_on && rp2_on && !k.res_patch && d->res && d->res == d->in && d->ldr == d->ldi && d->ci == d->co_fill && d->co == d->co_fill &&
            d->ci % 32 == 0 && d->dtype == UPS_BF16 && !small && d->dact && k.dact_bits && d->dact_kind == UPS_ACT_LRELU && d->act_slope > 0.f &&
            (d->ldd & 63) == 0 && (((uintptr_t)k.dact_bits) & 7) == 0 && !d->mask_bits && !d->mask_grad && !d->d2s && !d->f8_deq && !d->out_f32 &&
            !d->res_act && d->act_in == UPS_ACT_NONE && dma_patch_on() && static_taps_on() && flip && !fwd)
            k.res_patch = 2;
    }
    if ((d->out_act || d->res_act) && (d->mask_grad || d->d2s)) return 1;
    // an fp8 copy of a post-activation output is the quantisation of the stored value: no second activation
    if (d->out_act && (d->out_f8 || d->out_f8_amax) && d->out_f8_act != UPS_ACT_NONE) return 1;
    k.tap_off = 0; k.tap_wi = 0;
    for (int t = 0; t < 9; ++t) {
        k.tap_off |= (unsigned long long)(((d->tap_dy[t] + 1) << 2) | (d->tap_dx[t] + 1)) << (4 * t);
        k.tap_wi |= (unsigned long long)d->tap_w[t] << (4 * t);
    }
    if (d->dtype == UPS_F16 && (d->f8_deq || d->in_f8 || d->out_f8 || d->out_f8_amax || d->mask_bits || d->mask_grad || d->d2s)) return 1;

    const int rc = (d->dtype == UPS_F32) ? launch_t<float>(k, s) : (d->dtype == UPS_F16 ? launch_t<f16>(k, s) : launch_t<bf16>(k, s));
    return rc == UPS_OK ? 0 : rc;
}
