#include "../../mirge_amd/csrc/kernels.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
using namespace mg;
template <int V> __device__ KEnt mk(const uint64_t* text, uint32_t p) {
    int64_t s0 = (int64_t)p - CTX_BACK;
    uint32_t st = s0 < 0 ? 0u : (uint32_t)s0;
    uint32_t wi = st >> 5; int bs = (int)(st & 31) * 2;
    uint64_t t0, t1, t2;
    if (V == 0) { t0 = text[wi]; t1 = text[wi + 1]; t2 = text[wi + 2]; }
    else if (V == 1) { t0 = __builtin_nontemporal_load(text + wi); t1 = __builtin_nontemporal_load(text + wi + 1); t2 = __builtin_nontemporal_load(text + wi + 2); }
    else { const volatile uint64_t* vt = text; t0 = vt[wi]; t1 = vt[wi + 1]; t2 = vt[wi + 2]; }
    uint64_t lo = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0;
    uint64_t hi = bs ? (t1 >> bs) | (t2 << (64 - bs)) : t1;
    if (V == 3 || s0 < 0) {
        int sh = s0 < 0 ? (int)(-s0) * 2 : 0;
        uint64_t carry = sh ? lo >> (64 - sh) : 0;
        hi = (hi << sh) | carry; lo = lo << sh;
    }
    KEnt e; e.pos = p; e.c0 = (uint32_t)lo; e.c1 = (uint32_t)(lo >> 32); e.c2 = (uint32_t)hi;
    return e;
}
template <int V> __global__ void k_all(const uint64_t* text, uint32_t T, KEnt* out) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) out[p] = mk<V>(text, p);
}
// same as V=0 but stores the three words separately
__global__ void k_sep(const uint64_t* text, uint32_t T, uint32_t* out) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) { KEnt e = mk<0>(text, p); out[p] = e.c0; out[T + p] = e.c1; out[2 * (size_t)T + p] = e.c2; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)
int main() {
    uint32_t T = 500000; size_t words = T / 32 + 10;
    std::vector<uint64_t> text(words, 0);
    srand(1); for (size_t i = 0; i < T / 32 + 1; ++i) text[i] = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ rand();
    uint64_t* d; KEnt* o; uint32_t* o2; CK(hipMalloc(&d, words * 8)); CK(hipMalloc(&o, (size_t)T * 16)); CK(hipMalloc(&o2, (size_t)T * 12));
    CK(hipMemcpy(d, text.data(), words * 8, hipMemcpyHostToDevice));
    std::vector<KEnt> ref(T); for (uint32_t p = 0; p < T; ++p) ref[p] = make_kent(text.data(), p);
    for (int v = 0; v < 5; ++v) for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(o, 0, (size_t)T * 16));
        if (v == 0) k_all<0><<<(T + 255) / 256, 256>>>(d, T, o);
        if (v == 1) k_all<1><<<(T + 255) / 256, 256>>>(d, T, o);
        if (v == 2) k_all<2><<<(T + 255) / 256, 256>>>(d, T, o);
        if (v == 3) k_all<3><<<(T + 255) / 256, 256>>>(d, T, o);
        if (v == 4) k_sep<<<(T + 255) / 256, 256>>>(d, T, o2);
        CK(hipDeviceSynchronize());
        long bad = 0;
        if (v < 4) { std::vector<KEnt> g(T); CK(hipMemcpy(g.data(), o, (size_t)T * 16, hipMemcpyDeviceToHost));
            for (uint32_t p = 0; p < T; ++p) bad += (ref[p].pos != g[p].pos || ref[p].c0 != g[p].c0 || ref[p].c1 != g[p].c1 || ref[p].c2 != g[p].c2); }
        else { std::vector<uint32_t> g((size_t)T * 3); CK(hipMemcpy(g.data(), o2, (size_t)T * 12, hipMemcpyDeviceToHost));
            for (uint32_t p = 0; p < T; ++p) bad += (ref[p].c0 != g[p] || ref[p].c1 != g[T + p] || ref[p].c2 != g[2 * (size_t)T + p]); }
        printf("variant %d rep %d: bad %ld\n", v, rep, bad);
    }
    return 0;
}
