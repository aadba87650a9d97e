// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950 (no ISA manual in the image): operand layout, scale semantics and rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_scale_probe.hip -o /tmp/mfma_scale_probe && /tmp/mfma_scale_probe
// Layout assumed (and checked with exact small-integer e4m3 data): lane l holds row / column l & 15 and the 32 consecutive
// k values 32 (l >> 4) .. + 31 of BOTH operands; C/D as every 16x16 MFMA (col = l & 15, row = 4 (l >> 4) + reg).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) long i64x2;

__global__ void one(const int* a, const int* b, float* c, int sa, int sb) {
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[threadIdx.x * 8 + i]; vb[i] = b[threadIdx.x * 8 + i]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) c[threadIdx.x * 4 + i] = acc[i];
}

// rate kernels: operands resident in registers, independent accumulators, nothing else in the loop
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8p;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE> __global__ void __launch_bounds__(256, 4) rate(const int* a, float* c, int iters) {
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[(threadIdx.x & 63) * 8 + i]; vb[i] = a[512 + (threadIdx.x & 63) * 8 + i]; }
    float s = 0;
    if constexpr (MODE == 3) {
        bf16x8p x, y; __builtin_memcpy(&x, &va, 16); __builtin_memcpy(&y, &vb, 16);
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[j], 0, 0, 0); asm volatile("" : "+v"(acc[j])); }
        }
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    } else {
        bf16x8p x, y; __builtin_memcpy(&x, &va, 16); __builtin_memcpy(&y, &vb, 16);
        const long x8 = ((long)va[1] << 32) | (unsigned)va[0], y8 = ((long)vb[1] << 32) | (unsigned)vb[0];
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (MODE == 0) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else if constexpr (MODE == 1) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x8, y8, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[j], 0, 0, 0);
                asm volatile("" : "+v"(acc[j]));      // accumulate in place (hipcc otherwise rotates the tiles through AGPR copies)
            }
        }
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
    if (s == 12345.f) c[0] = s;
}

static float e4m3(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    const unsigned char vals[8] = {0x00, 0x38, 0x40, 0xB8, 0xC0, 0x30, 0x44, 0xB0};      // 0 1 2 -1 -2 .5 3 -.5
    std::vector<unsigned char> A(16 * 128), B(16 * 128);
    srand(1);
    for (auto& x : A) x = vals[rand() & 7];
    for (auto& x : B) x = vals[rand() & 7];
    std::vector<int> ha(64 * 8), hb(64 * 8);
    for (int l = 0; l < 64; ++l) { memcpy(&ha[l * 8], &A[(l & 15) * 128 + 32 * (l >> 4)], 32); memcpy(&hb[l * 8], &B[(l & 15) * 128 + 32 * (l >> 4)], 32); }
    int *da, *db; float* dc;
    hipMalloc(&da, 4096); hipMalloc(&db, 4096); hipMalloc(&dc, 1024);
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
    const int scales[4][2] = {{0x7f7f7f7f, 0x7f7f7f7f}, {0, 0}, {(int)0x80808080u, 0x7f7f7f7f}, {0x7f7f7f7f, 0x7e7e7e7e}};
    for (int t = 0; t < 4; ++t) {
        one<<<1, 64>>>(da, db, dc, scales[t][0], scales[t][1]);
        std::vector<float> hc(256);
        hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost);
        double maxd = 0, ratio = 0; int nr = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            double ref = 0; for (int k = 0; k < 128; ++k) ref += (double)e4m3(A[row * 128 + k]) * e4m3(B[col * 128 + k]);
            maxd = fmax(maxd, fabs(ref - hc[l * 4 + r]));
            if (fabs(ref) > 1) { ratio += hc[l * 4 + r] / ref; ++nr; }
        }
        printf("scale_a %08x scale_b %08x: max |D - A B^T| = %g, mean D / ref = %g\n", scales[t][0], scales[t][1], maxd, ratio / nr);
    }
    // rate: operands resident in registers, 8 independent accumulators per wave, 4 waves per SIMD, every CU busy.  The figure is
    // FLOP per cycle x the clock the chip HOLDS under that load (power): random operands draw more than small integers.
    for (int pass = 0; pass < 3; ++pass) {
    if (pass == 2) { std::fill(ha.begin(), ha.end(), 0); std::fill(hb.begin(), hb.end(), 0); printf("-- all-zero operands\n"); }
    else if (pass == 1) {        // random bit patterns with sane exponents: bf16 values ~N(0,1) (as fp8 bytes: a mix of magnitudes)
        std::vector<unsigned short> r(2048);
        for (auto& x : r) { float f = ((rand() & 0xffff) / 32768.f - 1.f) * 1.7f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
        for (int i = 0; i < 2048; i += 2) { r[i] = (r[i] & 0x7f7f) | (rand() & 0x8080); }      // keep fp8 bytes finite (no NaN codes)
        memcpy(ha.data(), r.data(), 2048); memcpy(hb.data(), r.data() + 1024, 2048);
        printf("-- random operands\n");
    } else printf("-- small-integer operands\n");
    hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(da + 512, hb.data(), 2048, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 4; wps *= 4) {
    const int iters = 20000 * (wps == 1 ? 4 : 1), blocks = 256 * wps;     // 256-thread blocks: one wave on each SIMD of a CU
    printf("   %d wave(s) per SIMD\n", wps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"scaled 16x16x128 e4m3", "16x16x32 fp8_fp8", "16x16x32 bf16", "32x32x16 bf16"};
    const double flop[4] = {2.0 * 16 * 16 * 128 * 8, 2.0 * 16 * 16 * 32 * 8, 2.0 * 16 * 16 * 32 * 8, 2.0 * 32 * 32 * 16 * 4};   // per wave and iteration
    const double cyc[4] = {32 * 8, 16 * 8, 16 * 8, 32 * 4};                                      // MFMA-pipe cycles per wave and iteration
    for (int m = 0; m < 4; ++m) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (m == 0) rate<0><<<blocks, 256>>>(da, dc, iters); else if (m == 1) rate<1><<<blocks, 256>>>(da, dc, iters);
        else if (m == 2) rate<2><<<blocks, 256>>>(da, dc, iters); else rate<3><<<blocks, 256>>>(da, dc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-24s %8.2f ms  %7.1f TFLOP/s   pipe-bound clock %.2f GHz\n", names[m], ms, flop[m] * iters * blocks * 4 / ms / 1e9,
                        cyc[m] * iters * wps / (ms * 1e6));
    }
    }
    }
    return 0;
}
