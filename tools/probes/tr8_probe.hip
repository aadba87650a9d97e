// Probe of ds_read_b64_tr_b8 on gfx950 (no ISA manual in the image): which LDS bytes does lane i of a 16-lane group receive, and
// what do the lanes' own addresses select?   hipcc --offload-arch=gfx950 -O3 tools/probes/tr8_probe.hip -o /tmp/tr8 && /tmp/tr8
// Hypothesis (by analogy with ds_read_b64_tr_b16, guide T10): the 16 lanes of a group address a block of 8 rows x 16 one-byte
// columns -- lane i points at row i >> 1, 8-byte piece i & 1 -- and lane i receives column i of the 8 rows (row 0 in byte 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) int v2i;
typedef __attribute__((address_space(3))) v2i lds_v2i;

__global__ void probe(unsigned long long* out, int rs, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = 0xee;
    __syncthreads();
    // rows r = 0..7 of group g at byte offset (g * 8 + r) * rs; row r holds bytes 16 r + c, c = 0..15 (+ 128 g would overflow:
    // groups are told apart by the row offset alone, every group holds the same 0..127 pattern)
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
        for (int r = 0; r < 8; ++r) if (i == 0) for (int c = 0; c < 16; ++c) sm[(g * 8 + r) * rs + c] = (unsigned char)(16 * r + c);
    }
    __syncthreads();
    const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
    int addr;
    if (mode == 0) addr = (g * 8 + (i >> 1)) * rs + (i & 1) * 8;          // hypothesis
    else if (mode == 1) addr = (g * 8 + (i & 7)) * rs + (i >> 3) * 8;      // alternative: row = i & 7, piece = i >> 3
    else addr = (g * 8) * rs + i * 8;                                      // contiguous 128 bytes (only meaningful for rs == 16)
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(sm + addr));
    out[threadIdx.x] = ((unsigned long long)(unsigned)r.y << 32) | (unsigned)r.x;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    unsigned long long h[64];
    const int rss[3] = {16, 64, 80};
    for (int mode = 0; mode < 3; ++mode)
        for (int k = 0; k < 3; ++k) {
            if (mode == 2 && rss[k] != 16) continue;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, rss[k], mode);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("mode %d row stride %d\n", mode, rss[k]);
            int ok = 1;
            for (int l = 0; l < 64; ++l) {
                unsigned char b[8]; memcpy(b, &h[l], 8);
                if (l < 16 || l % 16 == 0) {
                    printf("  lane %2d:", l);
                    for (int j = 0; j < 8; ++j) printf(" %3d", b[j]);
                    printf("\n");
                }
                for (int j = 0; j < 8; ++j) if (b[j] != 16 * j + (l & 15)) ok = 0;
            }
            printf("  => lane i receives column i of rows 0..7 (byte j = row j): %s\n", ok ? "YES" : "no");
        }
    return 0;
}
