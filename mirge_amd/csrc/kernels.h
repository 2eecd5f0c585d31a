// kernels.h -- the HIP kernels of the hot path (gfx950).  No reference kernel exists for any of
// them: the reference does this work in Perl hashes (miRge.pl:291-319, 453-494) and inside the
// external bowtie process (miRge.pl:327-339, 353).  Integer/bit work throughout; HBM- and
// latency-bound, no MFMA.
#pragma once
#include "keys.h"

namespace mg {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

// One reference library on the device.
//   text : 2-bit, 32 nt per word; the references back to back with ONE separator nt between
//          (and after) them; >= 8 zero words of padding at the end.
//   bad  : 1 bit per nt, set on reference N (non-ACGT) and on separators; a window is alignable
//          iff it holds no set bit, which also keeps it inside one reference.
//   koff/kpos : CSR k-mer index over every text position p: bucket = code of text[p, p+kb),
//          first nt most significant, so a shorter prefix is a contiguous bucket range.
struct LibDev {
    const uint64_t* text;
    const uint64_t* bad;
    const uint32_t* koff;
    const uint32_t* kpos;
    const uint32_t* ref_start;   // n_refs + 1 text coordinates
    uint32_t n_refs;
    uint32_t T;                  // text length in nt, separators included
    int32_t  kb;
};

struct RoundDev { int32_t lib, len_lt, len_gt, trim5, trim3, seedlen, seedmm, maxmm; };

// ------------------------------------------------------------------------------------------
// pack: ASCII matrix -> 2-bit keys (two streams).  counters: [0] n16, [1] n48, [2] error code,
// [3] first offending read.
// 16 bytes (chunk c of a row, as four little-endian words) -> 2-bit codes / N mask
MG_HD void pack_chunk(const uint32_t (&wv)[4], int c, int len, uint64_t (&s)[4], uint64_t (&nm)[2], bool& hasN, bool& bad) {
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        int pos = c * 16 + b;
        uint8_t ch = (uint8_t)(wv[b >> 2] >> (8 * (b & 3)));
        int code = base_code(ch);
        bool in = pos < len;
        bad |= in && code < 0;
        bool isN = in && code == 4;
        hasN |= isN;
        uint64_t cc = (in && code >= 0 && code < 4) ? (uint64_t)code : 0ull;
        s[c >> 1] |= cc << (2 * (pos & 31));
        nm[c >> 2] |= (uint64_t)isN << (pos & 63);
    }
}

__global__ void __launch_bounds__(256)
k_pack(const uint8_t* __restrict__ seqs, uint32_t width, const uint8_t* __restrict__ lens, uint64_t n,
       uint64_t* __restrict__ k16, uint64_t* __restrict__ k48, unsigned long long* counters) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int len = lens[i];
    if (len > 124 || (uint32_t)len > width) {
        if (atomicCAS(&counters[2], 0ull, 5ull) == 0ull) counters[3] = i;
        return;
    }
    uint64_t s[4] = {0, 0, 0, 0}, nm[2] = {0, 0};
    bool hasN = false, bad = false;
    const uint4* row = reinterpret_cast<const uint4*>(seqs + i * (uint64_t)width);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c * 16 < len) {
            uint4 v = row[c];
            uint32_t wv[4] = {v.x, v.y, v.z, v.w};
            pack_chunk(wv, c, len, s, nm, hasN, bad);
        }
    }
    if (bad) {
        if (atomicCAS(&counters[2], 0ull, 4ull) == 0ull) counters[3] = i;
        return;
    }
    if (!hasN && len <= 56) {
        unsigned long long p = atomicAdd(&counters[0], 1ull);
        k16[2 * p] = s[0];
        k16[2 * p + 1] = s[1] | ((uint64_t)len << 56);
    } else {
        unsigned long long p = atomicAdd(&counters[1], 1ull);
        k48[6 * p + 0] = s[0]; k48[6 * p + 1] = s[1]; k48[6 * p + 2] = s[2];
        k48[6 * p + 3] = s[3] | ((uint64_t)len << 56);
        k48[6 * p + 4] = nm[0]; k48[6 * p + 5] = nm[1];
    }
}

// ------------------------------------------------------------------------------------------
// collapse, step A: duplicates inside a tile are folded in LDS (miRNA-seq is dominated by a few
// sequences: one global atomic per tile instead of one per read), then every tile-distinct key
// is inserted into a call-local open-addressing table whose slots hold the READ INDEX of the
// first key that claimed them -- the keys array is immutable during the kernel, so comparing
// against keys[slot] needs no inter-workgroup hand-off.  cnt[rep read] accumulates the count.
template <int KW, int T, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_collapse_local(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* lslots, uint32_t lmask,
                 uint32_t* cnt, uint32_t* rep_out) {
    __shared__ uint64_t tk[T * KW];
    __shared__ uint32_t towner[2 * T];
    __shared__ uint32_t tcnt[T];
    __shared__ uint32_t tglob[T];
    constexpr int PER = T / BLOCK;
    const uint32_t base = blockIdx.x * (uint32_t)T;
    const uint32_t tid = threadIdx.x;
    const uint32_t nvalid = (n - base < (uint32_t)T) ? n - base : (uint32_t)T;
    for (uint32_t j = tid; j < (uint32_t)(T * KW); j += BLOCK)
        tk[j] = (j < nvalid * KW) ? keys[(uint64_t)base * KW + j] : 0ull;
    for (uint32_t j = tid; j < 2u * T; j += BLOCK) towner[j] = EMPTY;
    for (uint32_t j = tid; j < (uint32_t)T; j += BLOCK) tcnt[j] = 0;
    __syncthreads();
    uint32_t trep[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        uint32_t idx = tid + e * BLOCK;
        trep[e] = EMPTY;
        if (idx < nvalid) {
            uint64_t k[KW];
#pragma unroll
            for (int w = 0; w < KW; ++w) k[w] = tk[idx * KW + w];
            uint32_t h = key_hash<KW>(k) & (2u * T - 1);
            for (;;) {
                uint32_t o = atomicCAS(&towner[h], EMPTY, idx);
                if (o == EMPTY) { trep[e] = idx; break; }
                if (key_eq<KW>(&tk[o * KW], k)) { trep[e] = o; break; }
                h = (h + 1) & (2u * T - 1);
            }
            atomicAdd(&tcnt[trep[e]], 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        uint32_t idx = tid + e * BLOCK;
        if (idx < nvalid && trep[e] == idx) {
            uint64_t k[KW];
#pragma unroll
            for (int w = 0; w < KW; ++w) k[w] = tk[idx * KW + w];
            uint32_t gi = base + idx;
            uint32_t h = (key_hash<KW>(k) >> 7) & lmask;       // different bits than the tile table used
            uint32_t g;
            for (;;) {
                uint32_t o = atomicCAS(&lslots[h], EMPTY, gi);
                if (o == EMPTY) { g = gi; break; }
                uint64_t ok[KW];
#pragma unroll
                for (int w = 0; w < KW; ++w) ok[w] = keys[(uint64_t)o * KW + w];
                if (key_eq<KW>(ok, k)) { g = o; break; }
                h = (h + 1) & lmask;
            }
            atomicAdd(&cnt[g], tcnt[idx]);
            tglob[idx] = g;
        }
    }
    if (rep_out) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            uint32_t idx = tid + e * BLOCK;
            if (idx < nvalid) rep_out[base + idx] = tglob[trep[e]];
        }
    }
}

// collapse, step B: every representative read (cnt > 0) becomes one (unique id, count) entry of
// this sample.  If a table of earlier samples' uniques exists (gslots != null) it is looked up
// first (read-only here); unseen keys are appended to ukeys at U0 + atomic rank.
// ctr: [0] = new uniques appended, [1] = entries emitted.  gid_of (nullable): unique id per
// representative read index.
// A single hot counter takes ~12 ns per atomic on this chip, so ranks are formed per 2048-element tile
// (block scan in LDS) and each tile costs one atomic per counter.
constexpr int EMIT_PER = 8;
template <int KW>
__global__ void __launch_bounds__(256)
k_collapse_emit(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cnt, uint32_t n,
                const uint32_t* __restrict__ gslots, uint32_t gmask, uint64_t* ukeys, uint32_t U0,
                uint32_t* coo_g, uint32_t* coo_c, unsigned long long* ctr, uint32_t* gid_of) {
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_base[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * (256u * EMIT_PER);
    uint32_t c[EMIT_PER], g[EMIT_PER];
    uint32_t nrep = 0, nnew = 0;
#pragma unroll
    for (int e = 0; e < EMIT_PER; ++e) {
        uint32_t i = base + e * 256 + tid;
        c[e] = i < n ? cnt[i] : 0u;
        g[e] = EMPTY;
        if (c[e] && gslots) {
            uint64_t k[KW];
#pragma unroll
            for (int w = 0; w < KW; ++w) k[w] = keys[(uint64_t)i * KW + w];
            uint32_t h = (key_hash<KW>(k) >> 7) & gmask;
            for (;;) {
                uint32_t o = gslots[h];
                if (o == EMPTY) break;
                uint64_t ok[KW];
#pragma unroll
                for (int w = 0; w < KW; ++w) ok[w] = ukeys[(uint64_t)o * KW + w];
                if (key_eq<KW>(ok, k)) { g[e] = o; break; }
                h = (h + 1) & gmask;
            }
        }
        nrep += c[e] != 0;
        nnew += (c[e] != 0) && g[e] == EMPTY;
    }
    // block exclusive scan of (nrep | nnew << 16)
    uint32_t mine = nrep | (nnew << 16);
    s_scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = (int)tid >= o ? s_scan[tid - o] : 0u;
        __syncthreads();
        s_scan[tid] += t;
        __syncthreads();
    }
    uint32_t incl = s_scan[tid];
    if (tid == 255) {
        uint32_t trep = incl & 0xFFFF, tnew = incl >> 16;
        s_base[0] = trep ? (uint32_t)atomicAdd(&ctr[1], (unsigned long long)trep) : 0u;
        s_base[1] = tnew ? (uint32_t)atomicAdd(&ctr[0], (unsigned long long)tnew) : 0u;
    }
    __syncthreads();
    uint32_t prep = s_base[0] + (incl & 0xFFFF) - nrep;
    uint32_t pnew = s_base[1] + (incl >> 16) - nnew;
#pragma unroll
    for (int e = 0; e < EMIT_PER; ++e) {
        if (!c[e]) continue;
        uint32_t i = base + e * 256 + tid;
        uint32_t gg = g[e];
        if (gg == EMPTY) {
            gg = U0 + pnew++;
#pragma unroll
            for (int w = 0; w < KW; ++w) ukeys[(uint64_t)gg * KW + w] = keys[(uint64_t)i * KW + w];
        }
        coo_g[prep] = gg;
        coo_c[prep] = c[e];
        ++prep;
        if (gid_of) gid_of[i] = gg;
    }
}

// collapse, step C: insert uniques [g0, g1) into the cross-sample table.  They are distinct from
// each other and from everything already there, so a bare claim suffices.
template <int KW>
__global__ void __launch_bounds__(256)
k_gtable_insert(const uint64_t* __restrict__ ukeys, uint32_t g0, uint32_t g1, uint32_t* gslots, uint32_t gmask) {
    uint32_t g = g0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= g1) return;
    uint64_t k[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) k[w] = ukeys[(uint64_t)g * KW + w];
    uint32_t h = (key_hash<KW>(k) >> 7) & gmask;
    for (;;) {
        if (atomicCAS(&gslots[h], EMPTY, g) == EMPTY) break;
        h = (h + 1) & gmask;
    }
}

// per-read unique id: uid[i] = gid_of[rep[i]]
__global__ void __launch_bounds__(256)
k_uid(const uint32_t* __restrict__ rep, const uint32_t* __restrict__ gid_of, uint32_t n, uint32_t* uid, uint32_t flag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) uid[i] = gid_of[rep[i]] | flag;
}

// ------------------------------------------------------------------------------------------
// k-mer seed index build (set-up time; the reference pays bowtie-build + an index load per round)
MG_HD uint32_t text_kmer(const uint64_t* __restrict__ text, uint32_t p, int kb) {
    uint32_t wi = p >> 5; int bs = (p & 31) * 2;
    uint64_t t0 = text[wi], t1 = text[wi + 1];
    uint64_t v = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0;
    v &= ones(2 * kb);
    return (uint32_t)(rev2(v) >> (64 - 2 * kb));
}
__global__ void __launch_bounds__(256)
k_kmer_hist(const uint64_t* __restrict__ text, uint32_t T, int kb, uint32_t* hist) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) atomicAdd(&hist[text_kmer(text, p, kb)], 1u);
}
__global__ void __launch_bounds__(256)
k_kmer_scatter(const uint64_t* __restrict__ text, uint32_t T, int kb, uint32_t* cursor, uint32_t* kpos) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) kpos[atomicAdd(&cursor[text_kmer(text, p, kb)], 1u)] = p;
}

// exclusive scan of uint32, three kernels: per-chunk sums, scan of the sums (one block), apply.
constexpr int SCAN_CHUNK = 4096;   // 256 threads x 16
__global__ void __launch_bounds__(256)
k_scan_sums(const uint32_t* __restrict__ in, uint64_t n, uint32_t* sums) {
    __shared__ uint32_t red[256];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    uint32_t s = 0;
    for (int j = 0; j < 16; ++j) { uint64_t i = base + threadIdx.x + j * 256; if (i < n) s += in[i]; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(1024)
k_scan_top(uint32_t* sums, uint32_t m) {          // in place, exclusive; one block
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < m; b += 1024) {
        uint32_t i = b + threadIdx.x;
        uint32_t v = i < m ? sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = buf[threadIdx.x], c = carry;
        if (i < m) sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
k_scan_apply(const uint32_t* __restrict__ in, uint64_t n, const uint32_t* __restrict__ sums, uint32_t* out) {
    __shared__ uint32_t buf[256];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * 16;   // 16 consecutive per thread
    uint32_t v[16], s = 0;
    for (int j = 0; j < 16; ++j) { uint64_t i = base + j; v[j] = i < n ? in[i] : 0; s += v[j]; }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + buf[threadIdx.x] - s;
    for (int j = 0; j < 16; ++j) { uint64_t i = base + j; if (i < n) out[i] = run; run += v[j]; }
}

// ------------------------------------------------------------------------------------------
// matching.  Policy = bowtie 1 `-n` mode as miRge.pl:327-339 drives it: after
// -5/-3 trimming, ungapped end-to-end, either strand, <= seedmm mismatches in the first
// min(seedlen, len) read bases, <= maxmm overall, no reference N; of all valid alignments the best
// stratum (fewest seed mismatches) is kept and its representative is the maximum by
// (total mismatches, text position, strand) -- hypothesis H1 for "the last line bowtie prints",
// which is the one miRge keeps (miRge.pl:429-430).
struct Best { int sm; uint64_t score; uint32_t minpos, maxpos; };

// One candidate window [w, w+l) on one strand: valid?  (mismatch counts out)
template <int SW, bool HASN>
MG_HD bool window_ok(const LibDev& L, const Oriented<SW>& q, const uint64_t (&nsp)[SW], const uint64_t (&seedm)[SW],
                     int nN, int nNseed, int64_t w, const RoundDev& P, int& sm_out, int& tm_out) {
    if (w < 0 || w + q.l > (int64_t)L.T) return false;
    uint32_t wi = (uint32_t)(w >> 5); int bs = (int)(w & 31) * 2;
    uint64_t t[SW + 1];
#pragma unroll
    for (int i = 0; i <= SW; ++i) t[i] = L.text[wi + i];
    int tm = nN, sm = nNseed;
#pragma unroll
    for (int i = 0; i < SW; ++i) {
        uint64_t x = bs ? (t[i] >> bs) | (t[i + 1] << (64 - bs)) : t[i];
        uint64_t d = x ^ q.s[i];
        d = (d | (d >> 1)) & M55 & ones(2 * q.l - 64 * i);
        if (HASN) d &= ~nsp[i];
        tm += popc64(d);
        sm += popc64(d & seedm[i]);
    }
    if (tm > P.maxmm || sm > P.seedmm) return false;
    // reference N / reference boundary inside the window?
    uint32_t bi = (uint32_t)(w >> 6); int sh = (int)(w & 63);
    uint64_t b0 = L.bad[bi], b1 = L.bad[bi + 1];
    uint64_t r0 = sh ? (b0 >> sh) | (b1 << (64 - sh)) : b0;
    uint64_t any = r0 & ones(q.l);
    if (SW > 2) {
        uint64_t b2 = L.bad[bi + 2];
        uint64_t r1 = sh ? (b1 >> sh) | (b2 << (64 - sh)) : b1;
        any |= r1 & ones(q.l - 64);
    }
    if (any) return false;
    sm_out = sm; tm_out = tm;
    return true;
}

// Three order keys of a valid alignment; 0 = none.  Larger stratum field = fewer seed mismatches.
//   score : (4-sm, tm, w, fw)  -> its maximum is the representative (hypothesis H1)
//   kmax  : (4-sm, w)          -> max position inside the best stratum
//   kmin  : (4-sm, ~w)         -> min position inside the best stratum
MG_HD uint64_t hit_score(int sm, int tm, uint32_t w, int fw) { return ((uint64_t)(4 - sm) << 36) | ((uint64_t)tm << 33) | ((uint64_t)w << 1) | (uint64_t)fw; }
MG_HD uint64_t hit_kmax(int sm, uint32_t w) { return ((uint64_t)(4 - sm) << 32) | w; }
MG_HD uint64_t hit_kmin(int sm, uint32_t w) { return ((uint64_t)(4 - sm) << 32) | (0xFFFFFFFFu - w); }

// 1-bit mask -> the same bits at even positions of two words
MG_HD uint64_t spread32(uint32_t v) {
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

MG_HD uint32_t ref_of(const LibDev& L, uint32_t pos) {
    uint32_t lo = 0, hi = L.n_refs;            // largest j with ref_start[j] <= pos
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (L.ref_start[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}

constexpr int MAXP = 4;                         // pieces per strand: maxmm + 1 <= 4

// A read prepared for one round: both orientations, seed length, pigeonhole pieces.
template <int SW> struct Query {
    Oriented<SW> q[2];                          // [1] forward, [0] reverse complement
    int l, s, span, np;
    bool ok;                                    // false: bowtie skips the read (empty / l <= seedmm)
};
template <int KW>
MG_HD void query_setup(const uint64_t* key, const RoundDev& P, Query<KeyTraits<KW>::SW>& Q) {
    constexpr int SW = KeyTraits<KW>::SW;
    key_to_read<KW>(key, P.trim5, P.trim3, Q.q[1]);
    const int l = Q.q[1].l;
    Q.l = l; Q.ok = !(l <= 0 || l <= P.seedmm);
    if (!Q.ok) { Q.s = 0; Q.span = 0; Q.np = 0; return; }
    revcomp<SW>(Q.q[1], Q.q[0]);
    const int s = P.seedlen < l ? P.seedlen : l;
    // pigeonhole: <= seedmm mismatches in the seed and <= maxmm overall, so one of (A) seedmm+1 pieces of
    // the seed or (B) maxmm+1 pieces of the whole read matches exactly; take the longer pieces
    const int nA = (P.seedmm < P.maxmm ? P.seedmm : P.maxmm) + 1, nB = P.maxmm + 1;
    Q.s = s; Q.span = s; Q.np = nA;
    if (l / nB > s / nA) { Q.span = l; Q.np = nB; }
}
// per-strand constants of the mismatch count
template <int SW, bool HASN> struct Strand { uint64_t seedm[SW], nsp[SW]; int nN, nNseed; bool ok; };
template <int SW, bool HASN>
MG_HD void strand_setup(const Oriented<SW>& o, int fw, int l, int s, const RoundDev& P, Strand<SW, HASN>& S) {
    range_mask<SW>(fw ? 0 : l - s, fw ? s : l, S.seedm);
    S.nN = 0; S.nNseed = 0; S.ok = true;
    if (HASN) {
#pragma unroll
        for (int i = 0; i < SW; ++i) S.nsp[i] = spread32((uint32_t)(o.n[i >> 1] >> (32 * (i & 1))));
#pragma unroll
        for (int i = 0; i < SW; ++i) { S.nN += popc64(S.nsp[i]); S.nNseed += popc64(S.nsp[i] & S.seedm[i]); }
        S.ok = !(S.nN > P.maxmm || S.nNseed > P.seedmm);
    } else {
#pragma unroll
        for (int i = 0; i < SW; ++i) S.nsp[i] = 0;
    }
}
// piece p of strand fw -> where it sits in the oriented read (qs) and its bucket range [b0, b1] in koff
template <int SW, bool HASN>
MG_HD bool piece_buckets(const Query<SW>& Q, const Oriented<SW>& o, int kb, int fw, int p, int& qs, uint64_t& b0, uint64_t& b1) {
    if (p >= Q.np) return false;
    int ps = Q.span * p / Q.np, pe = Q.span * (p + 1) / Q.np;
    qs = fw ? ps : Q.l - pe;
    int kq = pe - ps < kb ? pe - ps : kb;
    if (kq <= 0) return false;
    if (HASN && n_in_range<SW>(o, qs, kq)) return false;      // a piece holding a read N is not the exact one
    uint32_t code = kmer_code<SW>(o, qs, kq);
    int sh = 2 * (kb - kq);
    b0 = (uint64_t)code << sh; b1 = ((uint64_t)code + 1) << sh;
    return true;
}

// Serial form of the matcher (one read, all candidates in turn): the definition the kernel below
// parallelises, and what tests/hostsim runs on the CPU.
template <int KW>
__host__ __device__ inline bool match_one(const uint64_t* key, const LibDev& L, const RoundDev& P, int32_t& ref, uint32_t& amb) {
    constexpr int SW = KeyTraits<KW>::SW;
    constexpr bool HASN = KeyTraits<KW>::HASN;
    Query<SW> Q;
    query_setup<KW>(key, P, Q);
    if (!Q.ok) return false;
    uint64_t score = 0, kmax = 0, kmin = 0;
    for (int fw = 1; fw >= 0; --fw) {
        const Oriented<SW>& o = Q.q[fw];
        Strand<SW, HASN> S;
        strand_setup<SW, HASN>(o, fw, Q.l, Q.s, P, S);
        if (!S.ok) continue;
        for (int p = 0; p < Q.np; ++p) {
            int qs; uint64_t b0, b1;
            if (!piece_buckets<SW, HASN>(Q, o, L.kb, fw, p, qs, b0, b1)) continue;
            uint32_t lo = L.koff[b0], hi = L.koff[b1];
            for (uint32_t c = lo; c < hi; ++c) {
                int64_t w = (int64_t)L.kpos[c] - qs; int sm, tm;
                if (window_ok<SW, HASN>(L, o, S.nsp, S.seedm, S.nN, S.nNseed, w, P, sm, tm)) {
                    uint64_t a = hit_score(sm, tm, (uint32_t)w, fw), b = hit_kmax(sm, (uint32_t)w), d = hit_kmin(sm, (uint32_t)w);
                    score = a > score ? a : score; kmax = b > kmax ? b : kmax; kmin = d > kmin ? d : kmin;
                }
            }
        }
    }
    if (!score) return false;
    ref = (int32_t)ref_of(L, (uint32_t)(score >> 1));
    amb = ref_of(L, (uint32_t)kmax) != ref_of(L, 0xFFFFFFFFu - (uint32_t)kmin);
    return true;
}

// One cascade round over the unique table: eligibility (miRge.pl:382-393), match, mark
// (miRge.pl:445-446).  stats: [0] processed, [1] aligned (miRge.pl:409-413).
//
// Candidate lists differ by orders of magnitude between reads (a 4-nt seed piece of a 13-nt isomiR
// query against the miRNA library has ~280 candidates, a 14-nt piece ~1), so a thread-per-read loop
// runs at the pace of the unluckiest lane.  The block therefore works in three phases:
//   1. every thread prepares its read, looks up all its piece ranges, verifies the short lists
//      (<= thr candidates) itself and queues the long ones in LDS;
//   2. the waves of the block drain the queue, one entry at a time, 64 candidates per step, the
//      owner's query being read back from LDS (same address in every lane: a broadcast);
//   3. every thread reads its result.
// A valid alignment is recorded by three LDS atomic max on its owner's keys (hit_score/kmax/kmin);
// valid alignments are rare next to candidates, so the atomics are off the critical path.
constexpr int MB = 256;
constexpr int STAT_SLOTS = 256;                  // per round: [processed | aligned] x STAT_SLOTS words, 64 B apart
constexpr int QCAP = 768;                        // LDS queue entries per block; a full queue falls back to the serial loop
template <int KW, int RN>
__global__ void __launch_bounds__(MB)
k_match_round(const uint64_t* __restrict__ ukeys, uint32_t U, uint8_t* ann_round, int32_t* ann_ref,
              uint8_t* ann_amb, LibDev L, RoundDev P, int thr, unsigned long long* stats) {
    constexpr int SW = KeyTraits<KW>::SW, NW = KeyTraits<KW>::NW;
    constexpr bool HASN = KeyTraits<KW>::HASN;
    constexpr int QW = 2 * SW + (HASN ? 2 * NW : 0);
    __shared__ unsigned long long s_best[3 * MB];
    __shared__ uint64_t s_q[MB * QW];
    __shared__ int s_ls[MB];
    __shared__ uint4 s_queue[QCAP];
    __shared__ uint32_t s_qn, s_next;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t u = blockIdx.x * MB + tid;
    bool active = false;
    uint64_t k[KW];
    if (u < U && ann_round[u] == 0) {
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = ukeys[(uint64_t)u * KW + w];
        int len = key_len<KW>(k);
        active = !(P.len_lt && !(len < P.len_lt)) && !(P.len_gt && !(len > P.len_gt));
    }
    if (!__syncthreads_or(active)) return;
    s_best[tid] = 0; s_best[MB + tid] = 0; s_best[2 * MB + tid] = 0;
    if (tid == 0) { s_qn = 0; s_next = 0; }
    __syncthreads();

    // ---- phase 1
    Query<SW> Q; Q.ok = false;
    if (active) query_setup<KW>(k, P, Q);
    if (Q.ok) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int i = 0; i < SW; ++i) s_q[tid * QW + f * SW + i] = Q.q[f].s[i];
            if (HASN) {
#pragma unroll
                for (int i = 0; i < NW; ++i) s_q[tid * QW + 2 * SW + f * NW + i] = Q.q[f].n[i];
            }
        }
        s_ls[tid] = Q.l | (Q.s << 8);
        int qs[2][MAXP]; uint32_t lo[2][MAXP], hi[2][MAXP];
        Strand<SW, HASN> S[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            strand_setup<SW, HASN>(Q.q[f], f, Q.l, Q.s, P, S[f]);
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                uint64_t b0 = 0, b1 = 0; qs[f][p] = 0;
                bool v = S[f].ok && piece_buckets<SW, HASN>(Q, Q.q[f], L.kb, f, p, qs[f][p], b0, b1);
                lo[f][p] = v ? L.koff[b0] : 0u;
                hi[f][p] = v ? L.koff[b1] : 0u;
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
#pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                uint32_t cnt = hi[f][p] - lo[f][p];
                if (cnt == 0) continue;
                uint32_t e = QCAP;
                if (cnt > (uint32_t)thr) { e = atomicAdd(&s_qn, 1u); }
                if (e < (uint32_t)QCAP) {
                    s_queue[e] = make_uint4(lo[f][p], hi[f][p], (uint32_t)qs[f][p] | ((uint32_t)f << 8) | (tid << 16), 0u);
                } else {
                    for (uint32_t c = lo[f][p]; c < hi[f][p]; ++c) {
                        int64_t w = (int64_t)L.kpos[c] - qs[f][p]; int sm, tm;
                        if (window_ok<SW, HASN>(L, Q.q[f], S[f].nsp, S[f].seedm, S[f].nN, S[f].nNseed, w, P, sm, tm)) {
                            atomicMax(&s_best[tid], (unsigned long long)hit_score(sm, tm, (uint32_t)w, f));
                            atomicMax(&s_best[MB + tid], (unsigned long long)hit_kmax(sm, (uint32_t)w));
                            atomicMax(&s_best[2 * MB + tid], (unsigned long long)hit_kmin(sm, (uint32_t)w));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: drain the queue, one entry per wave at a time
    const uint32_t qn = s_qn < (uint32_t)QCAP ? s_qn : (uint32_t)QCAP;
    for (;;) {
        uint32_t e = 0;
        if (lane == 0) e = atomicAdd(&s_next, 1u);
        e = __shfl(e, 0);
        if (e >= qn) break;
        const uint4 en = s_queue[e];
        const int eqs = (int)(en.z & 0xFF), f = (int)((en.z >> 8) & 1); const uint32_t owner = en.z >> 16;
        Oriented<SW> o;
#pragma unroll
        for (int i = 0; i < SW; ++i) o.s[i] = s_q[owner * QW + f * SW + i];
#pragma unroll
        for (int i = 0; i < NW; ++i) o.n[i] = HASN ? s_q[owner * QW + 2 * SW + f * NW + i] : 0ull;
        const int ls = s_ls[owner];
        o.l = ls & 0xFF;
        Strand<SW, HASN> S;
        strand_setup<SW, HASN>(o, f, o.l, ls >> 8, P, S);
        for (uint32_t c = en.x + lane; c < en.y; c += 64) {
            int64_t w = (int64_t)L.kpos[c] - eqs; int sm, tm;
            if (window_ok<SW, HASN>(L, o, S.nsp, S.seedm, S.nN, S.nNseed, w, P, sm, tm)) {
                atomicMax(&s_best[owner], (unsigned long long)hit_score(sm, tm, (uint32_t)w, f));
                atomicMax(&s_best[MB + owner], (unsigned long long)hit_kmax(sm, (uint32_t)w));
                atomicMax(&s_best[2 * MB + owner], (unsigned long long)hit_kmin(sm, (uint32_t)w));
            }
        }
    }
    __syncthreads();

    // ---- phase 3
    bool hit = false;
    if (Q.ok) {
        unsigned long long score = s_best[tid];
        if (score) {
            hit = true;
            ann_round[u] = (uint8_t)(RN + 1);
            ann_ref[u] = (int32_t)ref_of(L, (uint32_t)(score >> 1));
            ann_amb[u] = (uint8_t)(ref_of(L, (uint32_t)s_best[MB + tid]) != ref_of(L, 0xFFFFFFFFu - (uint32_t)s_best[2 * MB + tid]));
        }
    }
    // counters are spread over STAT_SLOTS cache lines: one hot word would cost ~12 ns per wave
    unsigned long long am = __ballot(active), hm = __ballot(hit);
    if (lane == 0) {
        uint32_t slot = (blockIdx.x * 4 + (tid >> 6)) & (STAT_SLOTS - 1);
        atomicAdd(&stats[(size_t)slot * 8], (unsigned long long)__popcll(am));
        if (hm) atomicAdd(&stats[(size_t)(STAT_SLOTS + slot) * 8], (unsigned long long)__popcll(hm));
    }
}

// ------------------------------------------------------------------------------------------
// count vectors (miRge.pl:465-493) over one sample's (unique id, count) entries.
// Per-block partial sums live in LDS (category totals, length histogram, and -- when the
// miRNA library is small enough -- the whole quant/iscan column), then one flush of atomics.
constexpr int CNT_LDS_M = 6144;
template <int KW>
__global__ void __launch_bounds__(256)
k_counts(const uint32_t* __restrict__ coo_g, const uint32_t* __restrict__ coo_c, uint32_t n, int sample, int S,
         const uint64_t* __restrict__ ukeys, const uint8_t* __restrict__ ann_round, const int32_t* __restrict__ ann_ref,
         int M, int round_mirna_a, int round_mirna_b, unsigned long long* quant, unsigned long long* iscan,
         unsigned long long* cat, unsigned long long* lenhist, unsigned long long* tuniq) {
    __shared__ unsigned long long s_cat[5];
    __shared__ unsigned long long s_len[128];
    __shared__ unsigned long long s_tu;
    __shared__ unsigned long long s_q[CNT_LDS_M];
    __shared__ unsigned long long s_i[CNT_LDS_M];
    const bool ldsm = M <= CNT_LDS_M;
    for (int j = threadIdx.x; j < 128; j += blockDim.x) s_len[j] = 0;
    if (threadIdx.x < 5) s_cat[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_tu = 0;
    if (ldsm) for (int j = threadIdx.x; j < M; j += blockDim.x) { s_q[j] = 0; s_i[j] = 0; }
    __syncthreads();
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        uint32_t g = coo_g[e];
        unsigned long long c = coo_c[e];
        int len = key_len<KW>(&ukeys[(uint64_t)g * KW]);
        int r = ann_round[g];
        atomicAdd(&s_tu, 1ull);                                      // miRge.pl:468
        atomicAdd(&s_len[len & 127], c);                             // miRge.pl:315
        int catidx;
        if (r == round_mirna_a || r == round_mirna_b) {              // miRge.pl:469
            catidx = 0;
            int32_t m = ann_ref[g];
            if (ldsm) { atomicAdd(&s_q[m], c); if (r == round_mirna_a) atomicAdd(&s_i[m], c); }
            else {
                atomicAdd(&quant[(uint64_t)m * S + sample], c);      // miRge.pl:473 / 476
                if (r == round_mirna_a) atomicAdd(&iscan[(uint64_t)m * S + sample], c);   // miRge.pl:474
            }
        } else if (r == 0) catidx = 4;                               // miRge.pl:489 remReads
        else catidx = (r >= 2 && r <= 4) ? r - 1 : 4;                // :480 hairpin, :483 other, :486 mRNA
        atomicAdd(&s_cat[catidx], c);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 128; j += blockDim.x) if (s_len[j]) atomicAdd(&lenhist[(uint64_t)j * S + sample], s_len[j]);
    if (threadIdx.x < 5 && s_cat[threadIdx.x]) atomicAdd(&cat[threadIdx.x * S + sample], s_cat[threadIdx.x]);
    if (threadIdx.x == 0 && s_tu) atomicAdd(&tuniq[sample], s_tu);
    if (ldsm) for (int j = threadIdx.x; j < M; j += blockDim.x) {
        if (s_q[j]) atomicAdd(&quant[(uint64_t)j * S + sample], s_q[j]);
        if (s_i[j]) atomicAdd(&iscan[(uint64_t)j * S + sample], s_i[j]);
    }
}

}  // namespace mg
