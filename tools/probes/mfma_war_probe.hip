// Probe: how long after a v_mfma_f32_16x16x32_bf16 does the hardware still READ its SrcC registers (gfx950)?
// hipcc rotates accumulators (D = A x B + C with D != C) and reuses the dead C registers for address arithmetic a few issue slots
// later, behind an s_nop of its own choosing; conv3x3_rows_kernel<bf16,64,6,10> came out irreproducible in exactly the lanes such a
// late SrcC read would explain (lanes 48..63 of C's first register; docs/design/negative_results.md).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_war_probe.hip -o /tmp/mfma_war && /tmp/mfma_war
//
// One asm block per case, physical registers, nothing for the compiler to schedule:  C = v[100:103] (a known pattern), A = 0, so the
// MFMA must return C exactly;  K issue slots after the MFMA a v_mov overwrites v100 with a NaN pattern.  Lanes of D's first register
// that come back NaN were read AFTER the overwrite.  Variants: the MFMA alone; followed by an independent MFMA (as in the kernel);
// preceded by an MFMA that produces its C (the dependent chain the kernel has).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define SETUP                                                                                                   \
    "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %4\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %4\n\t"                  \
    "v_mov_b32 v108, %4\n\tv_mov_b32 v109, %4\n\tv_mov_b32 v110, %4\n\tv_mov_b32 v111, %4\n\t"                  \
    "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"                      \
    "v_mov_b32 v116, %5\n\tv_mov_b32 v117, %5\n\tv_mov_b32 v118, %5\n\tv_mov_b32 v119, %5\n\t"                  \
    "v_mov_b32 v120, %5\n\tv_mov_b32 v121, %5\n\tv_mov_b32 v122, %5\n\tv_mov_b32 v123, %5\n\t"                  \
    "s_nop 7\n\t"
#define MFMA_T "v_mfma_f32_16x16x32_bf16 v[104:107], v[112:115], v[116:119], v[100:103]\n\t"      /* the one under test: D != C */
#define MFMA_I "v_mfma_f32_16x16x32_bf16 v[108:111], v[112:115], v[116:119], v[108:111]\n\t"      /* an independent one behind it */
#define MFMA_P "v_mfma_f32_16x16x32_bf16 v[100:103], v[112:115], v[116:119], v[100:103]\n\t"      /* a producer of its C in front */
#define WRITE "v_mov_b32 v100, 0x7fc00000\n\t"
#define WRITE3 "v_mov_b32 v103, 0x7fc00000\n\t"          /* the LAST register of C */
#define WRITEB "v_mov_b32 v116, 0x7fc07fc0\n\tv_mov_b32 v119, 0x7fc07fc0\n\t"      /* first and last register of SrcB */
/* read-after-write between MFMAs through the accumulator (A = B = 1.0 bf16 over K = 32: the producer adds 32) */
#define MFMA_P1T "v_mfma_f32_16x16x32_bf16 v[100:103], v[120:123], v[116:119], v[100:103]\n\t"     /* producer, tied: C += 32 */
#define MFMA_P1N "v_mfma_f32_16x16x32_bf16 v[104:107], v[120:123], v[116:119], v[100:103]\n\t"     /* producer, D != C: D = C + 32 */
#define MFMA_CN "v_mfma_f32_16x16x32_bf16 v[104:107], v[112:115], v[116:119], v[100:103]\n\t"      /* consumer of v[100:103], D != C, A = 0 */
#define MFMA_CT "v_mfma_f32_16x16x32_bf16 v[104:107], v[112:115], v[116:119], v[104:107]\n\t"      /* consumer of v[104:107], tied, A = 0 */
#define FINISH "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106\n\tv_mov_b32 %3, v107"
/* does the VALU's OWN write survive?  read back C's last / first register after the overwrite */
#define FINISH_C "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v103\n\tv_mov_b32 %1, v100\n\tv_mov_b32 %2, v104\n\tv_mov_b32 %3, v107"
#define WRITE03 "v_mov_b32 v103, 0x7fc00000\n\tv_mov_b32 v100, 0x7fc00000\n\t"
/* the epilogue's first arithmetic: VALU writes into C's registers (the residual halves), then a packed add with swapped halves */
#define PKADD "v_mov_b32 v103, %4\n\tv_mov_b32 v102, %5\n\tv_mov_b32 v122, %4\n\tv_mov_b32 v123, %4\n\t" \
              "v_pk_add_f32 v[120:121], v[122:123], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
#define FINISH_PK "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %0, v120\n\tv_mov_b32 %1, v121\n\tv_mov_b32 %2, v104\n\tv_mov_b32 %3, v107"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123"

#define NOPS0 ""
#define NOPS1 "s_nop 0\n\t"
#define NOPS2 "s_nop 1\n\t"
#define NOPS3 "s_nop 2\n\t"
#define NOPS4 "s_nop 3\n\t"
#define NOPS5 "s_nop 4\n\t"
#define NOPS6 "s_nop 5\n\t"
#define NOPS7 "s_nop 6\n\t"
#define NOPS8 "s_nop 7\n\t"
#define NOPS10 "s_nop 9\n\t"
#define NOPS12 "s_nop 11\n\t"
#define NOPS16 "s_nop 15\n\t"
#define NOPS20 "s_nop 15\n\ts_nop 3\n\t"

template <int VAR, int K>
__device__ __forceinline__ void one(float cval, unsigned bval, float& o0, float& o1, float& o2, float& o3) {
#define BODY(N)                                                                                                                         \
    if constexpr (VAR == 0) asm volatile(SETUP MFMA_T N WRITE FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);           \
    if constexpr (VAR == 1) asm volatile(SETUP MFMA_T MFMA_I N WRITE FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);    \
    if constexpr (VAR == 2) asm volatile(SETUP MFMA_P MFMA_I MFMA_T MFMA_I N WRITE FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);  \
    if constexpr (VAR == 7) asm volatile(SETUP MFMA_T N WRITE3 FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);          \
    if constexpr (VAR == 8) asm volatile(SETUP MFMA_T MFMA_I N WRITE3 FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);   \
    if constexpr (VAR == 9) asm volatile(SETUP MFMA_P MFMA_I MFMA_T N WRITEB FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB); \
    if constexpr (VAR == 12) asm volatile(SETUP MFMA_P MFMA_I MFMA_T MFMA_I N PKADD FINISH_PK : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB); \
    if constexpr (VAR == 10) asm volatile(SETUP MFMA_T N WRITE03 FINISH_C : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);         \
    if constexpr (VAR == 11) asm volatile(SETUP MFMA_P MFMA_I MFMA_T MFMA_I N WRITE03 FINISH_C : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB); \
    if constexpr (VAR == 3) asm volatile(SETUP MFMA_P1T N MFMA_CN FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);                  \
    if constexpr (VAR == 4) asm volatile(SETUP MFMA_P1T MFMA_I N MFMA_CN FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);           \
    if constexpr (VAR == 5) asm volatile(SETUP MFMA_P1N N MFMA_CT FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);                  \
    if constexpr (VAR == 6) asm volatile(SETUP MFMA_P1N MFMA_I N MFMA_CT FINISH : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(cval), "v"(bval) : CLOB);
    if constexpr (K == 0) { BODY(NOPS0) }
    if constexpr (K == 1) { BODY(NOPS1) }
    if constexpr (K == 2) { BODY(NOPS2) }
    if constexpr (K == 3) { BODY(NOPS3) }
    if constexpr (K == 4) { BODY(NOPS4) }
    if constexpr (K == 5) { BODY(NOPS5) }
    if constexpr (K == 6) { BODY(NOPS6) }
    if constexpr (K == 7) { BODY(NOPS7) }
    if constexpr (K == 8) { BODY(NOPS8) }
    if constexpr (K == 10) { BODY(NOPS10) }
    if constexpr (K == 12) { BODY(NOPS12) }
    if constexpr (K == 16) { BODY(NOPS16) }
    if constexpr (K == 20) { BODY(NOPS20) }
}

template <int VAR, int K>
__global__ __launch_bounds__(256) void probe(unsigned long long* bad, int iters) {
    const int lane = threadIdx.x & 63;
    bool w0 = false, wx = false;
    for (int it = 0; it < iters; ++it) {
        const float cval = 1.0f + (float)((it * 7 + lane) & 255);
        float o0, o1, o2, o3;
        one<VAR, K>(cval, 0x3f803f80u, o0, o1, o2, o3);
        const float want = (VAR >= 3 && VAR <= 6) ? cval + 32.0f : cval;     // (A = 0: D must equal C; VAR >= 3: C plus the producer's 32)
        if (it == 0 && blockIdx.x == 0 && threadIdx.x == 5) { ((float*)bad)[6] = o0; ((float*)bad)[7] = want; }
        if constexpr (VAR == 12) {      // o0 = v122 + v103 = 2 cval, o1 = v123 + v102 = cval + float(bval bits); o2 / o3 = D = C as it was
            w0 |= o0 != cval + cval || o1 != cval + __uint_as_float(0x3f803f80u);
            wx |= o2 != cval || o3 != cval;
        } else if constexpr (VAR >= 10) {      // o0 / o1 = C's last / first register after the VALU wrote the marker, o2 / o3 = D
            w0 |= __float_as_uint(o0) != 0x7fc00000u || __float_as_uint(o1) != 0x7fc00000u;
            wx |= o2 != cval || o3 != cval;
        } else {
            w0 |= o0 != want;
            wx |= o1 != want || o2 != want || o3 != want;          // the registers that are not overwritten
        }
    }
    const unsigned long long b0 = __ballot(w0), bx = __ballot(wx);
    if (lane == 0 && (b0 | bx)) { atomicOr(&bad[0], b0); atomicAdd(&bad[1], 1ull); atomicOr(&bad[2], bx); }
}

template <int VAR, int K>
static void run(unsigned long long* d, const char* what) {
    CK(hipMemset(d, 0, 32));
    hipLaunchKernelGGL((probe<VAR, K>), dim3(1024), dim3(256), 0, 0, d, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[4]; CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("%-52s %2d slots between the MFMA(s) and the write: wrong lanes of C's first register %016llx (others %016llx) in %llu of 4096 waves  [sample got %g want %g]\n", what, K, h[0], h[2], h[1], ((float*)h)[6], ((float*)h)[7]);
}

template <int VAR>
static void sweep(unsigned long long* d, const char* what) {
    run<VAR, 0>(d, what); run<VAR, 1>(d, what); run<VAR, 2>(d, what); run<VAR, 3>(d, what); run<VAR, 4>(d, what); run<VAR, 5>(d, what);
    run<VAR, 6>(d, what); run<VAR, 7>(d, what); run<VAR, 8>(d, what); run<VAR, 10>(d, what); run<VAR, 12>(d, what); run<VAR, 16>(d, what);
    run<VAR, 20>(d, what);
}

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 32));
    sweep<0>(d, "MFMA (D != C) alone");
    sweep<1>(d, "MFMA (D != C) + an independent MFMA");
    sweep<2>(d, "producer MFMA, indep., MFMA (D != C), indep.");
    // read after write through the accumulator: K slots between producer (and the independent MFMA behind it) and consumer
    sweep<7>(d, "MFMA (D != C) alone, LAST register of C overwritten");
    sweep<8>(d, "MFMA (D != C) + indep. MFMA, LAST register of C");
    sweep<9>(d, "chain of 3 MFMAs, then SrcB overwritten (A = 0)");
    sweep<12>(d, "chain + MFMAs in flight, VALU writes into C, v_pk_add_f32 op_sel");
    sweep<10>(d, "MFMA (D != C), then VALU writes C: does the WRITE survive");
    sweep<11>(d, "chain + MFMA (D != C) + indep., VALU writes C: survives?");
    sweep<3>(d, "RAW: tied producer -> consumer with D != C");
    sweep<4>(d, "RAW: tied producer, indep. MFMA -> consumer D != C");
    sweep<5>(d, "RAW: producer with D != C -> tied consumer");
    sweep<6>(d, "RAW: producer D != C, indep. MFMA -> tied consumer");
    return 0;
}
