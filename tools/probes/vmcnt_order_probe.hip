// Probe: do a wave's stores and its (slow) loads retire on vmcnt IN ISSUE ORDER on gfx950?  Every counted wait of the row-stream,
// prior and unpool kernels that leaves STORES in flight behind an LDS-DMA request relies on it (MI355X_MICROARCH.md, "vmcnt"; LLVM's
// SIInsertWaitcnts treats gfx9 loads and stores as one in-order event class).  If a young store could retire before an older load,
// `s_waitcnt vmcnt(K)` with K stores behind the load would pass with the load still in flight.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/vmcnt_order_probe.hip -o /tmp/vmcnt_probe && /tmp/vmcnt_probe
//
// form A: sentinel into the wave's LDS slot -> one global_load_lds_dwordx4 from a random 1 KiB chunk of a 4 GiB table (HBM and TLB
//         miss) -> K global_store_dword to the wave's own hot line (L2 hit) -> s_waitcnt vmcnt(K) -> ds_read the slot and compare.
// form B: the same with a global_load_dword into a VGPR holding the sentinel.
// form C: form A with NO wait at all (the control: shows that the probe can see a load in flight).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned tab(unsigned long long i) { return (unsigned)(i * 2654435761ull) ^ (unsigned)(i >> 11) ^ 0x5bd1e995u; }

__global__ void fill(unsigned* t, unsigned long long n) {
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) t[i] = tab(i);
}

#define ST1 "global_store_dword %3, %4, off\n\t"
#define ST2 ST1 ST1
#define ST4 ST2 ST2

template <int K, int FORM>
__global__ __launch_bounds__(256) void probe(const unsigned* table, unsigned chunks, unsigned* hot, unsigned long long* res, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) unsigned sm[4][256];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned rng = seed ^ (blockIdx.x * 2654435761u + wid * 40503u);
    unsigned long long stale = 0, wrong = 0;
    unsigned* hp = hot + ((size_t)blockIdx.x * 4 + wid) * 64 + lane;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)&sm[wid][0]);
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        const unsigned chunk = __builtin_amdgcn_readfirstlane((rng >> 4) % chunks);
        const unsigned* src = table + (size_t)chunk * 256;
        const unsigned val = rng;
        if constexpr (FORM == 0 || FORM == 2) {
            *(uint4*)&sm[wid][lane * 4] = make_uint4(0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned voff = lane * 16;
            if constexpr (FORM == 0) {
                if constexpr (K == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" ST1 "s_waitcnt vmcnt(1)" :: "s"(dst), "v"(voff), "s"(src), "v"(hp), "v"(val) : "memory", "m0");
                if constexpr (K == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" ST2 "s_waitcnt vmcnt(2)" :: "s"(dst), "v"(voff), "s"(src), "v"(hp), "v"(val) : "memory", "m0");
                if constexpr (K == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" ST4 "s_waitcnt vmcnt(4)" :: "s"(dst), "v"(voff), "s"(src), "v"(hp), "v"(val) : "memory", "m0");
            } else {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t" ST1 :: "s"(dst), "v"(voff), "s"(src), "v"(hp), "v"(val) : "memory", "m0");
            }
            const uint4 r = *(const uint4*)&sm[wid][lane * 4];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long b = (unsigned long long)chunk * 256 + lane * 4;
            const unsigned got[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (got[j] == tab(b + j)) continue;
                if (got[j] == 0xDEADBEEFu) ++stale; else ++wrong;
            }
        } else {
            unsigned r, o;
            const unsigned* lp = src + lane;
            if constexpr (K == 1) asm volatile("v_mov_b32 %0, 0xDEADBEEF\n\tglobal_load_dword %0, %2, off\n\t" ST1 "s_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)" : "=&v"(r), "=&v"(o) : "v"(lp), "v"(hp), "v"(val) : "memory");
            if constexpr (K == 2) asm volatile("v_mov_b32 %0, 0xDEADBEEF\n\tglobal_load_dword %0, %2, off\n\t" ST2 "s_waitcnt vmcnt(2)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)" : "=&v"(r), "=&v"(o) : "v"(lp), "v"(hp), "v"(val) : "memory");
            if constexpr (K == 4) asm volatile("v_mov_b32 %0, 0xDEADBEEF\n\tglobal_load_dword %0, %2, off\n\t" ST4 "s_waitcnt vmcnt(4)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)" : "=&v"(r), "=&v"(o) : "v"(lp), "v"(hp), "v"(val) : "memory");
            const unsigned long long b = (unsigned long long)chunk * 256 + lane;
            if (o != tab(b)) { if (o == 0xDEADBEEFu) ++stale; else ++wrong; }
            if (r != tab(b)) ++wrong;            // after vmcnt(0) the register must hold the table word
        }
    }
    if (stale) atomicAdd(&res[0], stale);
    if (wrong) atomicAdd(&res[1], wrong);
}

template <int K, int FORM>
static void run(const char* what, const unsigned* table, unsigned chunks, unsigned* hot, unsigned long long* res, int blocks, int iters) {
    CK(hipMemset(res, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL((probe<K, FORM>), dim3(blocks), dim3(256), 0, 0, table, chunks, hot, res, iters, 0x9e3779b9u * (rep + 1));
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, res, 16, hipMemcpyDeviceToHost));
    const double words = 4.0 * blocks * 4 * (double)iters * 64 * (FORM == 1 ? 1 : 4);
    printf("%-64s K=%d: %.3g words checked, %llu still the sentinel, %llu other mismatches  (%.1f ms)\n", what, K, words, h[0], h[1], ms);
}

int main() {
    const unsigned long long words = 1ull << 30;          // 4 GiB table
    unsigned* table; CK(hipMalloc(&table, words * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, table, words);
    const int blocks = 2048, iters = 2000;
    unsigned* hot; CK(hipMalloc(&hot, (size_t)blocks * 4 * 64 * 4));
    unsigned long long* res; CK(hipMalloc(&res, 16));
    CK(hipDeviceSynchronize());
    const unsigned chunks = (unsigned)(words / 256);
    run<1, 2>("control: LDS-DMA, one store, NO wait before the ds_read", table, chunks, hot, res, blocks, iters);
    run<1, 0>("LDS-DMA (HBM miss), K stores (L2 hit), vmcnt(K), ds_read", table, chunks, hot, res, blocks, iters);
    run<2, 0>("LDS-DMA (HBM miss), K stores (L2 hit), vmcnt(K), ds_read", table, chunks, hot, res, blocks, iters);
    run<4, 0>("LDS-DMA (HBM miss), K stores (L2 hit), vmcnt(K), ds_read", table, chunks, hot, res, blocks, iters);
    run<1, 1>("global_load_dword (HBM miss), K stores, vmcnt(K), v_mov", table, chunks, hot, res, blocks, iters);
    run<2, 1>("global_load_dword (HBM miss), K stores, vmcnt(K), v_mov", table, chunks, hot, res, blocks, iters);
    run<4, 1>("global_load_dword (HBM miss), K stores, vmcnt(K), v_mov", table, chunks, hot, res, blocks, iters);
    return 0;
}
