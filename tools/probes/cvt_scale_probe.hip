// Probe of v_cvt_scalef32_pk_{fp8,bf8}_bf16 on gfx950 (no ISA manual in the image): is the result src * scale or src / scale, which
// bits of the f32 scale count (its exponent only?), does an out-of-range value saturate to the largest normal or become inf / NaN,
// and is the rounding round-to-nearest-even?   hipcc --offload-arch=gfx950 -O3 tools/probes/cvt_scale_probe.hip -o /tmp/cvtp && /tmp/cvtp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

__global__ void k(const unsigned* in, unsigned* out, float sc, int n) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    bf16x2 v; __builtin_memcpy(&v, &in[i], 4);
    s16x2 z = {0, 0};
    s16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(z, v, sc, false);
    s16x2 b = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(z, v, sc, false);
    s16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(z, v, sc, true);
    unsigned ua, ub, uc; __builtin_memcpy(&ua, &a, 4); __builtin_memcpy(&ub, &b, 4); __builtin_memcpy(&uc, &c, 4);
    out[3 * i] = ua; out[3 * i + 1] = ub; out[3 * i + 2] = uc;
}

static float bf8(unsigned char v) {   // e5m2
    int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    float r = e == 0 ? ldexpf(m / 4.f, -14) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf(1.f + m / 4.f, e - 15));
    return s ? -r : r;
}
static float fp8(unsigned char v) {   // e4m3fn
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r = e == 0 ? ldexpf(m / 8.f, -6) : ((e == 15 && m == 7) ? NAN : ldexpf(1.f + m / 8.f, e - 7));
    return s ? -r : r;
}
static unsigned short tobf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    const float vals[] = {1.0f, 1.125f, 1.375f, 1.625f, 1.875f, 3.0f, 100.0f, 448.0f, 480.0f, 1000.0f, 57344.0f, 61440.0f, 1e6f, 1e-3f, -2.5f, 0.0f};
    const int n = sizeof(vals) / sizeof(float);
    unsigned h[n]; for (int i = 0; i < n; ++i) h[i] = tobf16(vals[i]) | ((unsigned)tobf16(-vals[i] * 0.5f) << 16);
    unsigned *din, *dout; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 12);
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    const float scales[] = {1.0f, 2.0f, 0.5f, 3.0f, 0.25f};
    for (float sc : scales) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, sc, n);
        unsigned o[n * 3]; hipMemcpy(o, dout, n * 12, hipMemcpyDeviceToHost);
        printf("scale %g:  src(lo, hi) -> bf8 bytes (lo, hi) = value | fp8 bytes = value | word_sel=1 result\n", sc);
        for (int i = 0; i < n; ++i)
            printf("  %10g %10g -> bf8 %02x %02x = %10g %10g | fp8 %02x %02x = %10g %10g | ws1 %08x\n", vals[i], -vals[i] * 0.5f,
                   o[3 * i] & 255, (o[3 * i] >> 8) & 255, bf8(o[3 * i] & 255), bf8((o[3 * i] >> 8) & 255),
                   o[3 * i + 1] & 255, (o[3 * i + 1] >> 8) & 255, fp8(o[3 * i + 1] & 255), fp8((o[3 * i + 1] >> 8) & 255), o[3 * i + 2]);
    }
    return 0;
}
