// keys.h -- 2-bit read keys and the bit arithmetic the matcher is built from.
//
// Key formats (the "coalesced HBM streams" of BASELINE.json's north_star):
//   16-byte key (KW = 2 words): reads of <= 56 nt made of A/C/G/T only.
//       w[0] bits 2i..2i+1 = nt i (i < 32); w[1] bits 0..47 = nt 32..55;
//       w[1] bits 56..63 = length.  A=0 C=1 G=2 T=3, so complement = ~code.
//   48-byte key (KW = 6 words): reads of <= 124 nt and/or holding N.
//       w[0..3] as above over 4 words (length in the top byte of w[3]);
//       w[4..5] = N mask, bit i set iff nt i is N (its 2-bit code is 0).
// The key is the exact sequence (miRge.pl:306 keys %seqHash by the string), so
// equality of keys == equality of reads.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define MG_HD __host__ __device__ __forceinline__

namespace mg {

constexpr uint64_t M55 = 0x5555555555555555ull;

template <int KW> struct KeyTraits;
template <> struct KeyTraits<2> { static constexpr int SW = 2, NW = 1, MAXLEN = 56;  static constexpr bool HASN = false; };
template <> struct KeyTraits<6> { static constexpr int SW = 4, NW = 2, MAXLEN = 124; static constexpr bool HASN = true;  };

template <int KW> MG_HD int key_len(const uint64_t* k) { return (int)(k[KeyTraits<KW>::SW - 1] >> 56); }

template <int KW> MG_HD uint32_t key_hash(const uint64_t* k) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < KW; ++i) { h ^= k[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return (uint32_t)h;
}
// All 64 bits of the same mix (key_hash is its low half).  The collapse tables take their slot from bits 16..47
// (slot_hash): the 25 bits that `key_hash >> 7` leaves gave at most 2^25 distinct starting slots, so a table of more
// than 33M uniques (a 100M-read shard) degenerated into probe chains millions of slots long.  The shard owner
// (shard.h) takes bits 41..63, independent of both.
template <int KW> MG_HD uint64_t key_hash64(const uint64_t* k) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < KW; ++i) { h ^= k[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return h;
}
MG_HD uint32_t slot_of_hash(uint64_t h64) { return (uint32_t)(h64 >> 16); }
template <int KW> MG_HD uint32_t slot_hash(const uint64_t* k) { return slot_of_hash(key_hash64<KW>(k)); }
template <int KW> MG_HD bool key_eq(const uint64_t* a, const uint64_t* b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < KW; ++i) e = e && (a[i] == b[i]);
    return e;
}

MG_HD uint64_t ones(int nbits) { return nbits >= 64 ? ~0ull : nbits <= 0 ? 0ull : ((1ull << nbits) - 1); }

// Reverse the order of the 32 2-bit groups of a word.
MG_HD uint64_t rev2(uint64_t x) {
    x = __builtin_bitreverse64(x);
    return ((x >> 1) & M55) | ((x & M55) << 1);
}

// word j of x, or 0 outside [0, W); j is a runtime value (select chain, stays in registers)
template <int W> MG_HD uint64_t pick(const uint64_t (&x)[W], int j) {
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) r = (i == j) ? x[i] : r;
    return r;
}
// 64 bits of the W-word little-endian integer x starting at bit bp (zeros beyond the end)
template <int W> MG_HD uint64_t bits_at(const uint64_t (&x)[W], int bp) {
    int wi = bp >> 6, bs = bp & 63;
    uint64_t lo = pick<W>(x, wi), hi = pick<W>(x, wi + 1);
    return bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
}
// out = x >> bits (logical, multiword)
template <int W> MG_HD void shr(const uint64_t (&x)[W], int bits, uint64_t (&out)[W]) {
#pragma unroll
    for (int i = 0; i < W; ++i) out[i] = bits_at<W>(x, bits + 64 * i);
}
// keep the low nbits of x
template <int W> MG_HD void keep_low(uint64_t (&x)[W], int nbits) {
#pragma unroll
    for (int i = 0; i < W; ++i) x[i] &= ones(nbits - 64 * i);
}

// A read in matching orientation: 2-bit sequence (nt 0 first), N mask, length.
template <int SW> struct Oriented {
    uint64_t s[SW];
    uint64_t n[(SW + 1) / 2];
    int l;
};

// Trimmed read (bowtie -5 a -3 b) out of a key.  l may come out <= 0.
template <int KW> MG_HD void key_to_read(const uint64_t* k, int trim5, int trim3, Oriented<KeyTraits<KW>::SW>& q) {
    constexpr int SW = KeyTraits<KW>::SW, NW = KeyTraits<KW>::NW;
    uint64_t s[SW];
#pragma unroll
    for (int i = 0; i < SW; ++i) s[i] = k[i];
    int len = (int)(s[SW - 1] >> 56);
    s[SW - 1] &= ones(56);
    int l = len - trim5 - trim3;
    q.l = l;
    if (l <= 0) l = 0;
    shr<SW>(s, 2 * trim5, q.s);
    keep_low<SW>(q.s, 2 * l);
    if (KeyTraits<KW>::HASN) {
        uint64_t n[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) n[i] = k[SW + i];
        shr<NW>(n, trim5, q.n);
        keep_low<NW>(q.n, l);
    } else {
#pragma unroll
        for (int i = 0; i < NW; ++i) q.n[i] = 0;
    }
}

// Reverse complement of an oriented read.
template <int SW> MG_HD void revcomp(const Oriented<SW>& a, Oriented<SW>& b) {
    constexpr int NW = (SW + 1) / 2;
    uint64_t t[SW];
#pragma unroll
    for (int i = 0; i < SW; ++i) t[i] = ~rev2(a.s[SW - 1 - i]);
    shr<SW>(t, 64 * SW - 2 * a.l, b.s);
    uint64_t m[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) m[i] = __builtin_bitreverse64(a.n[NW - 1 - i]);
    shr<NW>(m, 64 * NW - a.l, b.n);
    b.l = a.l;
}

// k-mer code of the kq nt starting at nt position p of an oriented read; first nt most significant,
// so every shorter prefix of a k-mer is a contiguous range of codes.
template <int SW> MG_HD uint32_t kmer_code(const Oriented<SW>& q, int p, int kq) {
    uint64_t v = bits_at<SW>(q.s, 2 * p) & ones(2 * kq);
    return (uint32_t)(rev2(v) >> (64 - 2 * kq));
}
// does the N mask of q have a bit in [p, p+kq) ?
template <int SW> MG_HD bool n_in_range(const Oriented<SW>& q, int p, int kq) {
    constexpr int NW = (SW + 1) / 2;
    return (bits_at<NW>(q.n, p) & ones(kq)) != 0;
}

// 1-bit-per-nt mask (at even bit positions) of nt range [a, b) inside a SW-word sequence
template <int SW> MG_HD void range_mask(int a, int b, uint64_t (&m)[SW]) {
#pragma unroll
    for (int i = 0; i < SW; ++i) m[i] = (ones(2 * b - 64 * i) ^ ones(2 * a - 64 * i)) & M55;
}

MG_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// ASCII -> code: A0 C1 G2 T3, N -> 4, anything else -> -1
MG_HD int base_code(uint8_t c) {
    int t = (c >> 1) & 3;                 // A0 C1 G3 T2
    int code = t ^ (t >> 1);              // A0 C1 G2 T3
    bool acgt = (c == 'A') | (c == 'C') | (c == 'G') | (c == 'T');
    return acgt ? code : (c == 'N' ? 4 : -1);
}

}  // namespace mg
