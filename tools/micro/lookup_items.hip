// Microbenchmark: what a "thread per lookup" kernel could reach -- 13M items (slot index, unique id), each thread
// fetches its item (coalesced), the 32-byte slot (random in a 537 MB table) and the unique's 16-byte key (ids ascend,
// so keys are near-sequential) at once, does ~100 integer ops on them, and 40 % of the items end in three 64-bit
// atomicMax on per-unique words.  To hold against round 3's ~0.5-0.6 ms for the same fetches inside the probe kernel.
// build: hipcc -O3 --offload-arch=gfx950 -o lookup_items lookup_items.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) k_make(uint32_t* item_slot, uint32_t* item_u, uint64_t n, uint64_t nslots, uint32_t U) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { item_slot[i] = (uint32_t)(mix(i + 17) % nslots); item_u[i] = (uint32_t)((i * (uint64_t)U) / n); }
}
template <int ATOM>
__global__ void __launch_bounds__(256, 6) k_items(const uint32_t* __restrict__ item_slot, const uint32_t* __restrict__ item_u, uint64_t n,
                                                   const u32x4* __restrict__ tab, const uint64_t* __restrict__ ukeys, unsigned long long* best) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = item_slot[i], u = item_u[i];
    const u32x4 a = tab[2 * (uint64_t)s], b = tab[2 * (uint64_t)s + 1];
    const uint64_t k0 = ukeys[2 * (uint64_t)u], k1 = ukeys[2 * (uint64_t)u + 1];
    // ~100 integer ops in the shape of the verification: align, xor, fold pairs, mask, popcount
    uint64_t c0 = (uint64_t)a.z | ((uint64_t)a.w << 32), c1 = (uint64_t)b.x | ((uint64_t)b.y << 32), c2 = (uint64_t)b.z | ((uint64_t)b.w << 32);
    const int sh = (int)(a.y & 62);
    uint64_t x0 = sh ? (c0 >> sh) | (c1 << (64 - sh)) : c0, x1 = sh ? (c1 >> sh) | (c2 << (64 - sh)) : c1;
    uint64_t d0 = x0 ^ k0, d1 = x1 ^ k1;
    d0 = (d0 | (d0 >> 1)) & 0x5555555555555555ull; d1 = (d1 | (d1 >> 1)) & 0x5555555555555555ull;
    int tm = __popcll(d0) + __popcll(d1), sm = __popcll(d0 & 0x0000FFFFFFFFFFFFull);
    for (int r = 0; r < 6; ++r) { d0 = mix(d0 + r) ; tm += (int)(d0 & 1); }
    const bool hitp = (mix(i) % 100) < 40;                       // 40 % of the items are alignments
    if (ATOM && hitp) {
        atomicMax(&best[3 * (uint64_t)u], (unsigned long long)(((uint64_t)(64 - sm) << 40) | ((uint64_t)tm << 33) | a.x));
        atomicMax(&best[3 * (uint64_t)u + 1], (unsigned long long)a.y);
        atomicMax(&best[3 * (uint64_t)u + 2], (unsigned long long)(0xFFFFFFFFu - a.y));
    } else if (((uint32_t)d0 ^ (uint32_t)tm ^ (uint32_t)sm) == 0x9E3779B9u) best[0] = d0;
}
int main() {
    const uint64_t n = 13000000ull, nslots = (537ull << 20) / 32; const uint32_t U = 10900000u;
    uint32_t *is, *iu; u32x4* tab; uint64_t* ukeys; unsigned long long* best;
    CK(hipMalloc(&is, n * 4)); CK(hipMalloc(&iu, n * 4)); CK(hipMalloc(&tab, nslots * 32)); CK(hipMalloc(&ukeys, (uint64_t)U * 16)); CK(hipMalloc(&best, (uint64_t)U * 24));
    CK(hipMemset(tab, 3, nslots * 32)); CK(hipMemset(ukeys, 5, (uint64_t)U * 16)); CK(hipMemset(best, 0, (uint64_t)U * 24));
    hipLaunchKernelGGL(k_make, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, is, iu, n, nslots, U);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int atom = 0; atom < 2; ++atom) {
        float bestms = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (atom) hipLaunchKernelGGL(k_items<1>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, is, iu, n, tab, ukeys, best);
            else hipLaunchKernelGGL(k_items<0>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, is, iu, n, tab, ukeys, best);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < bestms) bestms = ms;
        }
        printf("%s: %.3f ms for %llu items = %.1f G items/s\n", atom ? "with the three atomics on 40 %" : "fetch + verify only", bestms, (unsigned long long)n, n / bestms / 1e6);
    }
    return 0;
}
