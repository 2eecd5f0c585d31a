// shard.h -- the one exchange step of the multi-GPU path: unique tables are merged across
// ranks by key ownership so that every sequence is annotated on exactly one rank.  That makes the numbers
// that are NOT sums over reads -- trimmedUniq (miRge.pl:468) and the per-round unique counts
// (miRge.pl:410-413) -- additive over ranks, and removes the redundant annotation of sequences that occur
// in several shards.  No reference kernel exists: the reference is a single process.
//
// Record formats (what travels through the all-to-all):
//   16-byte class: 24 B = key[2] | sample u32 | count u32
//   48-byte class: 56 B = key[6] | sample u32 | count u32
// owner(key, world) = floor((key_hash64(key) >> 41) * world / 2^23): the top 23 bits, which the collapse tables do
// not use, so the keys a rank owns still spread over its whole table.
#pragma once
#include "keys.h"

namespace mg {

constexpr int SHARD_MAX_WORLD = 64;

template <int KW> MG_HD uint32_t key_owner(const uint64_t* k, uint32_t world) {
    return (uint32_t)(((key_hash64<KW>(k) >> 41) * world) >> 23);
}

#ifdef __HIPCC__
// entries per owner.  counts[world] accumulates over calls (one call per sample segment).
template <int KW>
__global__ void __launch_bounds__(256)
k_shard_count(const uint32_t* __restrict__ coo_g, uint32_t n, const uint64_t* __restrict__ ukeys, uint32_t world,
              unsigned long long* counts) {
    __shared__ uint32_t s_cnt[SHARD_MAX_WORLD];
    if (threadIdx.x < SHARD_MAX_WORLD) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) atomicAdd(&s_cnt[key_owner<KW>(&ukeys[(uint64_t)coo_g[e] * KW], world)], 1u);
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
}

// records grouped by owner: cursor[d] = next free record index of owner d's range (one atomic per block and owner)
template <int KW>
__global__ void __launch_bounds__(256)
k_shard_scatter(const uint32_t* __restrict__ coo_g, const uint32_t* __restrict__ coo_c, uint32_t n, uint32_t sample,
                const uint64_t* __restrict__ ukeys, uint32_t world, unsigned long long* cursor, uint64_t* __restrict__ rec) {
    __shared__ uint32_t s_cnt[SHARD_MAX_WORLD];
    __shared__ unsigned long long s_base[SHARD_MAX_WORLD];
    if (threadIdx.x < SHARD_MAX_WORLD) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t d = 0, r = 0, g = 0;
    if (e < n) { g = coo_g[e]; d = key_owner<KW>(&ukeys[(uint64_t)g * KW], world); r = atomicAdd(&s_cnt[d], 1u); }
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (e < n) {
        uint64_t* o = rec + (s_base[d] + r) * (KW + 1);
#pragma unroll
        for (int w = 0; w < KW; ++w) o[w] = ukeys[(uint64_t)g * KW + w];
        o[KW] = (uint64_t)sample | ((uint64_t)coo_c[e] << 32);
    }
}

// received records -> per-sample groups of (key, weight): count pass and scatter pass, per-block histograms in
// LDS (a single hot global counter costs ~12 ns per atomic; with one sample every record would hit the same one).
constexpr int REC_LDS_S = 2048;           // samples handled by this path; more is rejected by the host
template <int KW>
__global__ void __launch_bounds__(256)
k_rec_count(const uint64_t* __restrict__ rec, uint64_t n, uint32_t S, unsigned long long* counts, unsigned long long* bad) {
    __shared__ uint32_t s_cnt[REC_LDS_S];
    for (uint32_t j = threadIdx.x; j < S; j += blockDim.x) s_cnt[j] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t s = (uint32_t)rec[i * (KW + 1) + KW];
        if (s >= S) atomicAdd(bad, 1ull); else atomicAdd(&s_cnt[s], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < S; j += blockDim.x) if (s_cnt[j]) atomicAdd(&counts[j], (unsigned long long)s_cnt[j]);
}
template <int KW>
__global__ void __launch_bounds__(256)
k_rec_scatter(const uint64_t* __restrict__ rec, uint64_t n, uint32_t S, unsigned long long* cursor,
              uint64_t* __restrict__ keys, uint32_t* __restrict__ wt) {
    __shared__ uint32_t s_cnt[REC_LDS_S];
    __shared__ unsigned long long s_base[REC_LDS_S];
    for (uint32_t j = threadIdx.x; j < S; j += blockDim.x) s_cnt[j] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = S, r = 0; uint64_t tail = 0;
    if (i < n) {
        tail = rec[i * (KW + 1) + KW];
        s = (uint32_t)tail;
        if (s < S) r = atomicAdd(&s_cnt[s], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < S; j += blockDim.x) if (s_cnt[j]) s_base[j] = atomicAdd(&cursor[j], (unsigned long long)s_cnt[j]);
    __syncthreads();
    if (s < S) {
        const unsigned long long p = s_base[s] + r;
#pragma unroll
        for (int w = 0; w < KW; ++w) keys[p * KW + w] = rec[i * (KW + 1) + w];
        wt[p] = (uint32_t)(tail >> 32);
    }
}
#endif

}  // namespace mg
