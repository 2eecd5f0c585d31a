// Microbenchmark: rate of independent random 32-byte fetches (one per thread) from tables of growing size --
// the "speed of light" of a slot-table lookup on this machine, to hold a cascade round's lookup rate against.
// build: hipcc -O3 --offload-arch=gfx950 -o random_lines random_lines.hip ; run: ./random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// PER lookups per thread, all issued before any is used
template <int PER>
__global__ void __launch_bounds__(256, 6) k_fetch(const u32x4* __restrict__ tab, uint64_t nslots, uint64_t n, uint32_t* out, uint64_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * PER >= n) return;
    u32x4 a[PER], b[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const uint64_t s = mix(seed + i * PER + j) % nslots; a[j] = tab[2 * s]; b[j] = tab[2 * s + 1]; }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) acc += a[j].x ^ a[j].w ^ b[j].y ^ b[j].z;
    if (acc == 0x12345u) out[i & 1023] = acc;
}
int main() {
    const uint64_t n = 40000000ull;
    uint32_t* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t sizes_mb[] = {8, 64, 512, 2048, 8192, 32768};
    for (uint64_t mb : sizes_mb) {
        const uint64_t bytes = mb << 20, nslots = bytes / 32;
        u32x4* tab; CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 1, bytes));
        for (int per = 1; per <= 4; per *= 2) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                const uint32_t grid = (uint32_t)((n / per + 255) / 256);
                CK(hipEventRecord(e0));
                if (per == 1) hipLaunchKernelGGL(k_fetch<1>, dim3(grid), dim3(256), 0, 0, tab, nslots, n, out, (uint64_t)rep * 977);
                if (per == 2) hipLaunchKernelGGL(k_fetch<2>, dim3(grid), dim3(256), 0, 0, tab, nslots, n, out, (uint64_t)rep * 977);
                if (per == 4) hipLaunchKernelGGL(k_fetch<4>, dim3(grid), dim3(256), 0, 0, tab, nslots, n, out, (uint64_t)rep * 977);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
            }
            printf("table %6llu MB  %d in flight/thread: %.3f ms for %llu fetches = %.1f G fetches/s\n", (unsigned long long)mb, per, best, (unsigned long long)n, n / best / 1e6);
            fflush(stdout);
        }
        CK(hipFree(tab));
    }
    return 0;
}
