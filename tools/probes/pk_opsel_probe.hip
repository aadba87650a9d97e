// Probe (round 6, docs/design/rows_hazard.md): does a packed-FP32 VALU instruction return a wrong result while ANOTHER wave of the same
// SIMD streams MFMAs?  The row-stream kernel's irreproducible element was traced (observation builds ab/d_*) to
//     v_pk_add_f32 v[158:159], v[158:159], v[170:171] op_sel:[0,1] op_sel_hi:[1,0]
// delivering   low = src0.lo + 0   instead of   src0.lo + src1.hi   in lanes 48..63, with both inputs verified right before and after, only
// in the waves that run their epilogue while their SIMD sibling is still inside its MFMA section.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_opsel_probe.hip -o ab/pk_opsel_probe && ab/pk_opsel_probe
//
// One block = 8 waves = two per SIMD.  Waves 4..7 (role H) hammer v_mfma_f32_16x16x32_bf16 on non-zero data; waves 0..3 (role V) run
// the victim sequence over and over and count lanes whose result differs from the exact value.  Variants of the victim sequence:
//   0  v_pk_add_f32 with op_sel:[0,1] op_sel_hi:[1,0]                 (the kernel's form)
//   1  v_pk_add_f32 plain (op_sel:[0,0] op_sel_hi:[1,1])
//   2  v_pk_mul_f32 plain
//   3  v_add_f32 x 2 (scalar control)
//   4  the kernel's neighbourhood: v_cmp / s_nop 1 / v_cndmask in front of the op_sel form
// and of the hammer: 0 none (control), 1 MFMA stream, 2 MFMA stream + LDS reads, 3 LDS-DMA stream (global_load_lds_dwordx4 from an
// L2-resident buffer, the row-stream kernels' fetch), 4 LDS-DMA + MFMA, 5 the VICTIM wave issues the LDS-DMA itself (siblings idle).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int VAR, int HAM>
__global__ __launch_bounds__(512) void probe(unsigned long long* bad, float* sink, int iters, const unsigned char* zsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) ((float*)smem)[i] = 0.25f + (float)(i & 63);
    __syncthreads();
    const unsigned smem_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned d_off = (unsigned)(lane * 16);
    auto dma = [&](int slot) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(smem_lds + 32768u + (unsigned)(wave * 8192 + (slot & 7) * 1024));
        const unsigned char* src = zsrc + (size_t)((slot * 7 + blockIdx.x) & 15) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(d_off), "s"(src) : "memory", "m0");
    };
    if (wave >= 4) {
        if (HAM == 0 || HAM == 5) return;
        if (HAM == 3) {
            for (int it = 0; it < iters * 2; ++it) {
                dma(2 * it); dma(2 * it + 1);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.01f * (float)((lane + i) & 15)); b[i] = (__bf16)(1.0f - 0.02f * (float)((lane * 3 + i) & 7)); }
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters * 6; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            if (HAM == 4 && (it & 7) == 0) { dma(it >> 2); dma((it >> 2) + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            if (HAM == 2) {
                const f32x4 t = *(const f32x4*)(smem + ((lane * 16 + it * 1024) & 32767 & ~15));
                a[0] = (__bf16)(t[0] * 1e-3f + 0.5f);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678f) sink[threadIdx.x] = c0[0];
        return;
    }
    // victim
    unsigned long long wrong_lo = 0, wrong_hi = 0;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 x, y, r;
        x[0] = 1.0f + (float)((it + lane) & 127); x[1] = 300.0f + (float)((it * 3 + lane) & 63);
        y[0] = 0.5f + (float)((it * 5 + lane) & 31); y[1] = 1000.0f + (float)((it * 7 + lane) & 255);
        float w0, w1;
        if (HAM == 5 && (it & 3) == 0) { dma(it >> 1); dma((it >> 1) + 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        if (VAR == 0) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
            w0 = x[0] + y[1]; w1 = x[1] + y[0];
        } else if (VAR == 1) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            w0 = x[0] + y[0]; w1 = x[1] + y[1];
        } else if (VAR == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            w0 = x[0] * y[0]; w1 = x[1] * y[1];
        } else if (VAR == 3) {
            float r0, r1;
            asm volatile("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(r0), "=&v"(r1) : "v"(x[0]), "v"(x[1]), "v"(y[1]), "v"(y[0]));
            r[0] = r0; r[1] = r1;
            w0 = x[0] + y[1]; w1 = x[1] + y[0];
        } else {
            // v_cmp ; s_nop 1 ; v_cndmask (writes y's registers) ; the op_sel add -- the instruction neighbourhood of the kernel's epilogue
            float y1n;
            asm volatile("v_mov_b32 v100, %3\n\tv_mov_b32 v101, %4\n\t"
                         "v_cmp_lt_f32 vcc, 0, v101\n\ts_nop 1\n\tv_cndmask_b32 v101, %5, v101, vcc\n\t"
                         "v_pk_add_f32 %0, %2, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_mov_b32 %1, v101"
                         : "=&v"(r), "=&v"(y1n) : "v"(x), "v"(y[0]), "v"(y[1]), "v"(y[1] * 5.0f) : "vcc", "v100", "v101");
            y[1] = y1n;
            w0 = x[0] + y[1]; w1 = x[1] + y[0];
        }
        const bool b0 = r[0] != w0, b1 = r[1] != w1;
        wrong_lo |= __ballot(b0); wrong_hi |= __ballot(b1);
        nbad += (b0 || b1) ? 1u : 0u;
        if ((b0 || b1) && sink[4096] == 0.f) { sink[4097] = r[0]; sink[4098] = w0; sink[4099] = r[1]; sink[4100] = w1; sink[4096] = 1.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && (wrong_lo | wrong_hi)) { atomicOr(&bad[0], wrong_lo); atomicOr(&bad[1], wrong_hi); atomicAdd(&bad[2], 1ull); }
    if (nbad) atomicAdd(&bad[3], (unsigned long long)nbad);
}

static unsigned char* g_zsrc;
template <int VAR, int HAM>
static void run(unsigned long long* d, float* sink, const char* what) {
    CK(hipMemset(d, 0, 64)); CK(hipMemset(sink, 0, 8192 * 4));
    CK(hipFuncSetAttribute((const void*)probe<VAR, HAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipLaunchKernelGGL((probe<VAR, HAM>), dim3(1024), dim3(512), 98304, 0, d, sink, 4000, (const unsigned char*)g_zsrc);
    CK(hipDeviceSynchronize());
    unsigned long long h[8]; float s[8];
    CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(s, sink + 4096, 32, hipMemcpyDeviceToHost));
    printf("%-44s hammer %d: wrong lanes of the LOW result %016llx, of the HIGH result %016llx, %llu of 4096 victim waves, %llu lane-results",
           what, HAM, h[0], h[1], h[2], h[3]);
    if (s[0] != 0.f) printf("  [sample: low got %g want %g, high got %g want %g]", s[1], s[2], s[3], s[4]);
    printf("\n");
}

int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 8192 * 4));
    CK(hipMalloc(&g_zsrc, 65536)); CK(hipMemset(g_zsrc, 0, 65536));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 3>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<0, 4>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<0, 5>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<1, 3>(d, sink, "v_pk_add_f32 plain");
        run<1, 4>(d, sink, "v_pk_add_f32 plain");
        run<1, 5>(d, sink, "v_pk_add_f32 plain");
        run<3, 3>(d, sink, "2 x v_add_f32 (scalar control)");
        run<3, 4>(d, sink, "2 x v_add_f32 (scalar control)");
        run<3, 5>(d, sink, "2 x v_add_f32 (scalar control)");
        run<4, 3>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
        run<4, 4>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
        run<4, 5>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
        run<0, 0>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<0, 1>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<0, 2>(d, sink, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]");
        run<1, 1>(d, sink, "v_pk_add_f32 plain");
        run<1, 2>(d, sink, "v_pk_add_f32 plain");
        run<2, 1>(d, sink, "v_pk_mul_f32 plain");
        run<2, 2>(d, sink, "v_pk_mul_f32 plain");
        run<3, 1>(d, sink, "2 x v_add_f32 (scalar control)");
        run<3, 2>(d, sink, "2 x v_add_f32 (scalar control)");
        run<4, 0>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
        run<4, 1>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
        run<4, 2>(d, sink, "v_cmp / s_nop 1 / v_cndmask / op_sel add");
    }
    return 0;
}
