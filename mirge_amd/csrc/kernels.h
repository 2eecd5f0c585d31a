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
//   koff/kent : CSR k-mer index over every text position p: bucket = code of text[p, p+kb),
//          first nt most significant, so a shorter prefix is a contiguous bucket range.  Each entry
//          carries, next to the position, the 48 nt of text around it (KEnt): a candidate is judged
//          from that one 16-byte load, and the text itself is read only for windows the context
//          does not cover.
// index entry: text position + context = 2-bit text[pos - CTX_BACK, pos - CTX_BACK + CTX_NT)
struct __attribute__((aligned(16))) KEnt { uint32_t pos, c0, c1, c2; };
constexpr int CTX_BACK = 16, CTX_NT = 48;

// Block schemes: seeds for reads too short to be cut into exact-lookup pieces.  The first 4B nt of the read are cut
// into four blocks of B nt; with <= m mismatches among them at least 4 - m blocks are exact, so the C(4, m) shapes
// that use 4 - m blocks (m = 1: 1110, 0111, 1101, 1011; m = 2: 1100, 1010, 1001, 0110, 0101, 0011) between them find
// every such alignment, each through ONE bucket of (4 - m) * B nt of specificity.  A 16-nt seed with one mismatch
// allowed thus makes four 12-nt lookups instead of two 8-nt range scans (~55 candidates each in a 3.6 Mnt library);
// a 13-nt isomiR query with two mismatches allowed six 6-nt lookups instead of three 4-nt ones (~280 each in the
// miRNA library).  One both-strand slot table per shape (TabDesc).
// Entry of a both-strand table: 32 bytes that decide a candidate on their own -- position, reference, how far the
// reference runs clean (no N, no boundary) to either side, and 96 nt of text around the window -- so that a lookup
// touches ONE line and nothing else (no text, no N bitmap, no reference offsets; those cost ~5 more cache misses per
// aligned read while entries carried 48 nt).
//   pos  : text position of the looked-up window; bit 31 (POS_REV): the entry serves the reverse strand
//   info : reference index (bits 0-19) | dl (20-25) | dr (26-31): how many nt the window may reach to the left of /
//          from its lookup position on without meeting an N or a reference boundary (capped at 63: reads of the
//          16-byte class are <= 56 nt)
//   c    : 96 nt of text, 2-bit, laid out so that a read whose lookup starts ps nt into it lines up at context nt
//          40 - ps: text[pos - 40, pos + 56) for a forward entry.
// A reverse entry is stored MIRRORED: its context is the reverse complement of text[pos + K - 56, pos + K + 40)
// (K = length of the looked-up window: the k-mer, or a scheme's 4B) and dl / dr are measured in that orientation,
// so that the FORWARD read is compared against it exactly as against a forward entry -- the matcher never builds a
// reverse complement, and the strand only shows in the reported position.
struct __attribute__((aligned(32))) PEnt { uint32_t pos, info, c[6]; };
// slot of a compact table: a == b: empty bucket; b == CS_SINGLE: the bucket's only entry is a (position | strand);
// else the bucket's entries are cpos[a, b)
struct __attribute__((aligned(8))) CSlot { uint32_t a, b; };
constexpr uint32_t CS_SINGLE = 0xFFFFFFFFu;
constexpr int PCTX_NT = 96, PCTX_BACK = 40;
constexpr uint32_t PENT_MAX_REFS = 1u << 20;
// A table (the contiguous one or one scheme's shape tables) inside the library's slot / occupancy buffers.
struct TabDesc {
    uint32_t slot_off;           // first slot (LibDev::slots); a scheme's shape i starts nbk * i further on
    uint32_t bits_off;           // first occupancy word (LibDev::sbits); shape i starts nbk / 32 * i further on
    uint32_t info;               // kbs (0-7) | m (8-11) | occupancy bits in use (12) | blocks (16-31; 0 = contiguous)
};
// A scheme's four blocks, 4 bits each, block 0 (the read's first nt) in the low bits.  Equal blocks (B, B, B, B) are the
// original schemes; unequal ones let a scheme's window be EXACTLY a short read's length: a 13-nt read under (4,3,3,3)
// makes six lookups keyed by 6-7 nt (~9-36 entries per bucket of the miRNA library) where (3,3,3,3) over its first 12 nt
// made six keyed by 6 nt (~36 each); a 15-nt read under (4,4,4,3) six keyed by 7-8 nt (~2-9 each).
MG_HD uint32_t blocks_eq(int B) { return (uint32_t)B * 0x1111u; }
MG_HD int blk(uint32_t bl, int j) { return (int)((bl >> (4 * j)) & 15u); }
MG_HD int blocks_W(uint32_t bl) { return blk(bl, 0) + blk(bl, 1) + blk(bl, 2) + blk(bl, 3); }
MG_HD int tab_kbs(const TabDesc& t) { return (int)(t.info & 255); }
MG_HD int tab_m(const TabDesc& t) { return (int)((t.info >> 8) & 15); }
MG_HD bool tab_bits(const TabDesc& t) { return (t.info >> 12) & 1; }
MG_HD uint32_t tab_blocks(const TabDesc& t) { return t.info >> 16; }
MG_HD int tab_W(const TabDesc& t) { return blocks_W(tab_blocks(t)); }
MG_HD uint32_t tab_info(int kbs, uint32_t blocks, int m, bool bits) { return (uint32_t)kbs | ((uint32_t)m << 8) | ((uint32_t)bits << 12) | (blocks << 16); }
constexpr int MAX_SCHEMES = 8;                   // per library
constexpr int ROUND_SCHEMES = 6;                 // per round
// the schemes one round may use, most specific first (round_schemes)
struct RoundSch { int32_t n; TabDesc t[ROUND_SCHEMES]; };
struct LibDev {
    const uint64_t* text;
    const uint64_t* bad;
    const uint32_t* koff;
    const KEnt*     kent;
    const uint64_t* kpre;        // kent order: the middle 32 nt of each entry's context, text[pos-12, pos+20) -- 8-byte pre-filter for long scans
    const uint32_t* kbits;       // 1 bit per bucket: non-empty?  (4^kb / 8 bytes: cache-resident where koff is not)
    const uint32_t* ref_start;   // n_refs + 1 text coordinates
    const uint32_t* ref_lut;     // ref_lut[s] = reference that holds text position s << lut_shift (T >> lut_shift) + 2 entries
    int32_t  lut_shift;
    uint32_t n_refs;
    uint32_t T;                  // text length in nt, separators included
    int32_t  kb;
    // Both-strand slot tables (slots == null: none).  Every text position p is entered twice, as (p, fwd) under its
    // k-mer and as (p, rev) under the reverse complement of its k-mer, so ONE bucket answers a piece for both strands;
    // and a table is direct-addressed, one 32-byte slot per bucket: a bucket with one entry -- the usual non-empty
    // case -- holds that entry itself (PEnt).  An exact-bucket lookup is then the occupancy bit (cache-resident) + one
    // random line, for both strands, instead of two dependent pairs koff -> kent and the text behind them.
    // Slot: pos = SLOT_EMPTY: nothing; pos = SLOT_LIST: info / c[0] = the bucket's range in `pent`; else the entry.
    // All tables of a library share these buffers.
    const PEnt*     slots;
    const uint32_t* sbits;
    const PEnt*     pent;        // entries of the buckets that hold more than one
    const uint64_t* ppre;        // pent order: the 32 nt of each entry's context from the lookup position on (pent_pre) -- 8-byte pre-filter of the list drain
    TabDesc   ctab;              // the contiguous table: 4^kb slots
    int32_t   nsch;
    TabDesc   sch[MAX_SCHEMES];  // the block schemes built so far
    // Compact both-strand tables (cslots == null: none) of a library whose 2-bit text and N bitmap fit the LDS (the
    // miRNA library: 18 KB + 9 KB).  Same buckets, same TabDesc layout (ctab / sch then describe THESE tables and `slots`
    // is null), but a slot is 8 bytes and an entry 4: text position | strand.  A candidate is verified against the text
    // the block staged into LDS, so the whole index of such a library is ~10 MB instead of ~300 MB of 32-byte
    // self-contained entries, and stays in the L2 / Infinity Cache (k_lds_round).
    const CSlot*    cslots;
    const uint32_t* cpos;        // entries of the buckets that hold more than one
    uint32_t  lds_tw, lds_bw;    // 64-bit words of text / of the N bitmap a block stages (0: not such a library)
    // the reference of a text position as a rank query: sep = bitmap of the separators alone (null / lds_sw = 0: the
    // library holds no N, so `bad` is that bitmap), rank[j] = separators before bitmap word j
    const uint64_t* sep;
    const uint32_t* rank;
    uint32_t  lds_sw;
};
constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu, SLOT_LIST = 0x7FFFFFFFu, POS_REV = 0x80000000u, POS_MASK = 0x7FFFFFFFu;

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
k_collapse_local(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ wt, uint32_t n, uint32_t* lslots, uint32_t lmask,
                 uint32_t* cnt, uint32_t* cnt0) {
    // wt (nullable): a count per input key (records merged from other ranks, shard.h); null = one read each
    __shared__ uint64_t tk[T * KW];
    __shared__ uint32_t towner[2 * T];
    __shared__ uint32_t tcnt[T];
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
            atomicAdd(&tcnt[trep[e]], wt ? wt[base + idx] : 1u);
        }
    }
    __syncthreads();
    uint32_t mine[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        uint32_t idx = tid + e * BLOCK;
        mine[e] = 0;
        if (idx < nvalid && trep[e] == idx) {
            uint64_t k[KW];
#pragma unroll
            for (int w = 0; w < KW; ++w) k[w] = tk[idx * KW + w];
            uint32_t gi = base + idx;
            uint32_t h = slot_hash<KW>(k) & lmask;             // different bits than the tile table used
            uint32_t g;
            bool claimed = false;
            for (;;) {
                // look before claiming: a taken slot costs a load, not an atomic (a stale "empty" only
                // means the compare-and-swap below decides)
                uint32_t o = __hip_atomic_load(&lslots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (o == EMPTY) {
                    o = atomicCAS(&lslots[h], EMPTY, gi);
                    if (o == EMPTY) { g = gi; claimed = true; break; }
                }
                uint64_t ok[KW];
#pragma unroll
                for (int w = 0; w < KW; ++w) ok[w] = keys[(uint64_t)o * KW + w];
                if (key_eq<KW>(ok, k)) { g = o; break; }
                h = (h + 1) & lmask;
            }
            if (claimed) mine[e] = tcnt[idx];                   // the claimant's own count: a plain store below
            else atomicAdd(&cnt[g], tcnt[idx]);
        }
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        uint32_t idx = tid + e * BLOCK;
        if (idx < nvalid) cnt0[base + idx] = mine[e];           // 0 for every read that is not a representative
    }
}

// collapse, step B: every representative read (cnt > 0) becomes one (unique id, count) entry of
// this sample.  If a table of earlier samples' uniques exists (gslots != null) it is looked up
// first (read-only here); unseen keys are appended to ukeys at U0 + atomic rank.
// ctr[0] = (new uniques appended << 32) | entries emitted.
// A single hot counter takes ~12 ns per atomic on this chip, so ranks are formed per 2048-element tile
// (block scan in LDS) and each tile costs one atomic.
constexpr int EMIT_PER = 8;
template <int KW>
__global__ void __launch_bounds__(256)
k_collapse_emit(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cnt0, uint32_t n,
                const uint32_t* __restrict__ gslots, uint32_t gmask, uint64_t* ukeys, uint32_t U0,
                uint32_t* coo_g, uint32_t* coo_c, unsigned long long* ctr) {
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_base[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * (256u * EMIT_PER);
    uint32_t c[EMIT_PER], g[EMIT_PER];
    uint32_t nrep = 0, nnew = 0;
#pragma unroll
    for (int e = 0; e < EMIT_PER; ++e) {
        uint32_t i = base + e * 256 + tid;
        c[e] = 0u;
        if (i < n) { uint32_t own = cnt0[i]; c[e] = own ? own + cnt[i] : 0u; }     // representative: own count + the duplicates' adds
        g[e] = EMPTY;
        if (c[e] && gslots) {
            uint64_t k[KW];
#pragma unroll
            for (int w = 0; w < KW; ++w) k[w] = keys[(uint64_t)i * KW + w];
            uint32_t h = slot_hash<KW>(k) & gmask;
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
        // both counters in ONE word (new uniques << 32 | entries; each stays below 2^32): one hot atomic per tile, not two
        const unsigned long long old = (trep | tnew) ? atomicAdd(&ctr[0], ((unsigned long long)tnew << 32) | trep) : 0ull;
        s_base[0] = (uint32_t)old;
        s_base[1] = (uint32_t)(old >> 32);
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
    }
}

// A caller-built key stream checked against the formats of keys.h (mirge_validate_keys_dev): length byte within the
// class's limit, no bits above the sequence, N mask within the length and a zero code under every N.  flag[0] = 1 +
// index of the first offender found (0 = none), by atomicMin on the index.
MG_HD uint64_t spread32_pair(uint64_t m, int half) {       // 32 mask bits (low or high half of m) -> even bit positions of a word
    uint64_t x = (uint32_t)(m >> (32 * half));
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull;
    return (x | (x << 1)) & 0x5555555555555555ull;
}
template <int KW>
__global__ void __launch_bounds__(256)
k_validate_keys(const uint64_t* __restrict__ keys, uint64_t n, unsigned long long* flag) {
    constexpr int SW = KeyTraits<KW>::SW, NW = KeyTraits<KW>::NW;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) k[w] = keys[i * KW + w];
    const int len = (int)(k[SW - 1] >> 56);
    bool bad = len > KeyTraits<KW>::MAXLEN;
    k[SW - 1] &= ones(56);
#pragma unroll
    for (int w = 0; w < SW; ++w) bad = bad || (k[w] & ~ones(2 * len - 64 * w)) != 0;
    if (KeyTraits<KW>::HASN) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint64_t nm = k[SW + w];
            bad = bad || (nm & ~ones(len - 64 * w)) != 0;
            // the two sequence words under this mask word must hold code 0 wherever it is set
            const uint64_t lo = spread32_pair(nm, 0), hi = spread32_pair(nm, 1);
            bad = bad || (k[2 * w] & (lo | (lo << 1))) != 0 || (2 * w + 1 < SW && (k[2 * w + 1] & (hi | (hi << 1))) != 0);
        }
    }
    if (bad) atomicMin(flag, (unsigned long long)i + 1);
}

// ------------------------------------------------------------------------------------------
// Partitioned collapse (16-byte keys; BASELINE.json's "LDS-staged" collapse in hash form).  The table form above pays
// one device-scope compare-and-swap and ~3 random lines per distinct key; at the bench's U/N = 0.55 that is what its
// 0.9 ms are made of.  Here the reads are split by hash bits in two streaming passes of small fan-out (16 .. 512 by input size, then 256)
// into partitions that fit the LDS, and each partition is folded by one workgroup:
//   k_part_pass : a tile of 2048 records -> duplicates folded in an LDS table (one record per tile and sequence: the
//                 heavy hitters of miRNA-seq never reach the next pass more than once per tile) -> the tile's distinct
//                 records ranked per bucket in LDS -> ONE global atomic per tile and bucket reserves their run in the
//                 bucket's region -> 16-byte scattered stores, which the L2 combines into whole lines;
//   k_part_fold : one final partition (<= PT_C2 records) in LDS -> folded -> its uniques appended to the table as
//                 k_collapse_emit does (one packed atomic per partition).
// Regions have fixed capacities; a pass that would overrun one raises `overflow` and the host runs the table form.
#ifndef MG_PT_TILE1
#define MG_PT_TILE1 1024
#endif
#ifndef MG_PT_BLOCK1
#define MG_PT_BLOCK1 512
#endif
#ifndef MG_PT_TILE2
#define MG_PT_TILE2 2048
#endif
#ifndef MG_PT_BLOCK2
#define MG_PT_BLOCK2 1024
#endif
#ifndef MG_PT_P2
#define MG_PT_P2 256
#endif
#ifndef MG_PT_C2
#define MG_PT_C2 2048
#endif
#ifndef MG_PT_EXP
#define MG_PT_EXP 0                              // phase elimination (timing only, results wrong): profiles/r03/collapse_experiments.txt
#endif
#ifndef MG_PT_FBLOCK
#define MG_PT_FBLOCK 1024
#endif
// records of a tile / threads of a block, first and second split pass; threads of a fold block.  The passes differ: every
// tile of the FIRST pass bumps the same P1 cursors, the second pass's tiles spread over P1 x 256 of them (k_part_pass).
constexpr int PT_TILE1 = MG_PT_TILE1, PT_BLOCK1 = MG_PT_BLOCK1, PT_TILE2 = MG_PT_TILE2, PT_BLOCK2 = MG_PT_BLOCK2, PT_FBLOCK = MG_PT_FBLOCK;
constexpr int PT_P2 = MG_PT_P2;                  // fan-out of pass 2 (pass 1: 16 .. 512, by input size)
constexpr int PT_C2 = MG_PT_C2;                  // records a final partition may hold
constexpr int LL_CSTRIDE = 16;                   // 8-byte words between the two length words of the length-class lists (a line each: atomics on one line are served one after the other)
constexpr int pt_log2(int x) { return x <= 1 ? 0 : 1 + pt_log2(x / 2); }
// Buckets by multiply-shift on the top 32 hash bits x, so that pass 1 can have ANY fan-out P1: x * P1 = bucket * 2^32 +
// rest; pass 1 takes the bucket, pass 2 the top log2(P2) bits of the rest (uniform inside a bucket, and far above the
// hash bits the LDS tables use).  sel = 0: pass 1, 1: pass 2.
MG_HD uint32_t pt_bucket(uint64_t h64, uint32_t p1, int sel) {
    const uint64_t x = (h64 >> 32) * (uint64_t)p1;
    return sel ? (uint32_t)x >> (32 - pt_log2(PT_P2)) : (uint32_t)(x >> 32);
}
template <int NB, int PT_TILE, int PT_BLOCK>
__global__ void __launch_bounds__(PT_BLOCK)
k_part_pass(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ wt_in, const unsigned long long* __restrict__ n_in, uint32_t n_max,
            uint32_t in_stride, uint32_t p1, int sel, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ wt_out, unsigned long long* cursor,
            uint32_t cap_out, unsigned long long* overflow, uint32_t gmul, uint32_t gdiv, uint32_t cs_in, uint32_t cs_out) {
    // blockIdx.y = input partition (its records start in_stride records further on; n_in[p * cs_in] of them, at most
    // n_max), blockIdx.x = tile: a block per tile (resident blocks that stride over the tiles with the next tile on its
    // way into registers were measured slower: blocks that come and go overlap better).  Keys AND counts of a tile are
    // fetched before the first barrier: a count fetched where it is added (inside the de-duplication loop) put a global
    // load into the middle of the tile's LDS phase -- 0.53 -> 0.47 ms for both passes.  wt_in (nullable) = a count per
    // input record.
    // What bounds a pass (profiles/r03/collapse_experiments.txt, phase elimination at 20M reads, first pass): loads +
    // LDS fill alone 0.114 ms of 0.223; the stores 0.078; the de-duplication 0.03; the cursor atomics nothing -- a block
    // holds its LDS while its loads arrive and while its stores drain, and the pass moves 660 MB at 70 % of the
    // machine's copy rate.  What the atomics DO cost, they cost as a chain:
    // * cursors cs_out 8-byte words apart (first pass: 512 B, the host's part_cstride()): atomics on one 128-B line are
    //   served one after the other, and 16 cursors to a line made the first pass 0.269 ms instead of 0.223;
    // * cursor groups (gmul = G > 1 in the first pass, gdiv = G in the second): every tile of the first pass bumps every
    //   one of its P1 cursors, ~12 ns each per address, so the pass cannot be shorter than tiles x 12 ns (0.117 ms at
    //   tiles of 2048 records, 0.234 at 1024, 0.47 at 512 -- measured 0.223 / 0.291 / 0.496).  With G cursors (and
    //   regions) per bucket, tile t appending to region (bucket, t mod G), the chain is G times shorter and the first
    //   pass can take SMALLER tiles, more of which fit a CU: 1024 records x 512 threads, four blocks per CU, G = 8:
    //   0.196 ms.  The second pass takes the G regions of a bucket as G input partitions (blockIdx.y = bucket * G +
    //   group) that share the bucket's output cursors; its own cursors are P1 x 256, no chain, and it keeps tiles of 2048
    //   records (0.195 ms; 0.212 at 1024: runs of 4 records per bucket and tile combine worse than runs of 8).
    constexpr int PT_TSLOTS = 2 * PT_TILE;                      // slots of the tile's LDS table
    __shared__ uint64_t tk[PT_TILE * 2];
    __shared__ uint32_t towner[PT_TSLOTS];
    __shared__ uint32_t tcnt[PT_TILE];
    __shared__ uint32_t hist[NB];
    uint32_t* gbase = towner;                                   // (dead by the time the bases are written)
    static_assert(NB <= PT_TSLOTS, "bucket bases lie over the table slots");
    constexpr int PER = PT_TILE / PT_BLOCK, KL = 2 * PER;
    const uint32_t tid = threadIdx.x, p = blockIdx.y, po = p / gdiv, grp = blockIdx.x % gmul;
    const uint32_t n = n_in ? (uint32_t)(n_in[(uint64_t)p * cs_in] < n_max ? n_in[(uint64_t)p * cs_in] : n_max) : n_max;
    const uint64_t base = (uint64_t)blockIdx.x * PT_TILE;
    if (base >= n) return;
    const uint32_t nvalid = n - base < (uint64_t)PT_TILE ? (uint32_t)(n - base) : (uint32_t)PT_TILE;
    const uint64_t* kin = keys_in + ((uint64_t)p * in_stride + base) * 2;
    const uint32_t* win = wt_in ? wt_in + (uint64_t)p * in_stride + base : nullptr;
    uint32_t wcur[PER];
    {
        uint64_t pk[KL];
#pragma unroll
        for (int e = 0; e < KL; ++e) { const uint32_t j = tid + e * PT_BLOCK; pk[e] = j < nvalid * 2 ? kin[j] : 0ull; }
#pragma unroll
        for (int e = 0; e < PER; ++e) { const uint32_t idx = tid + e * PT_BLOCK; wcur[e] = (win && idx < nvalid) ? win[idx] : 1u; }
        for (uint32_t j = tid; j < (uint32_t)PT_TSLOTS; j += PT_BLOCK) towner[j] = EMPTY;
        for (uint32_t j = tid; j < (uint32_t)PT_TILE; j += PT_BLOCK) tcnt[j] = 0;
        for (uint32_t b = tid; b < (uint32_t)NB; b += PT_BLOCK) hist[b] = 0;
#pragma unroll
        for (int e = 0; e < KL; ++e) tk[tid + e * PT_BLOCK] = pk[e];
    }
    __syncthreads();
    uint32_t trep[PER]; uint64_t h64[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t idx = tid + e * PT_BLOCK;
        trep[e] = EMPTY; h64[e] = 0;
        if (idx < nvalid) {
            uint64_t k[2] = {tk[idx * 2], tk[idx * 2 + 1]};
            h64[e] = key_hash64<2>(k);
            uint32_t h = (uint32_t)h64[e] & (uint32_t)(PT_TSLOTS - 1);                             // (the buckets take the top 32 hash bits)
#if MG_PT_EXP == 3
            trep[e] = idx; tcnt[idx] = wcur[e]; (void)h;
#else
            for (;;) {
                const uint32_t o = atomicCAS(&towner[h], EMPTY, idx);
                if (o == EMPTY) { trep[e] = idx; break; }
                if (tk[o * 2] == k[0] && tk[o * 2 + 1] == k[1]) { trep[e] = o; break; }
                h = (h + 1) & (uint32_t)(PT_TSLOTS - 1);
            }
            atomicAdd(&tcnt[trep[e]], wcur[e]);
#endif
        }
    }
    __syncthreads();
#if MG_PT_EXP == 4
    if (tk[tid] == 0x1234567ull) keys_out[0] = 1;
    return;
#endif
    uint32_t rank[PER], bk[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t idx = tid + e * PT_BLOCK;
        rank[e] = EMPTY; bk[e] = 0;
        if (idx < nvalid && trep[e] == idx) { bk[e] = pt_bucket(h64[e], p1, sel); rank[e] = atomicAdd(&hist[bk[e]], 1u); }
    }
    __syncthreads();
    for (uint32_t b = tid; b < (uint32_t)NB; b += PT_BLOCK) {
        const uint32_t c = hist[b];
        unsigned long long at = 0;
        if (c) {
#if MG_PT_EXP == 2
            at = ((uint64_t)blockIdx.x * 40) % (cap_out - 100);
#else
            at = atomicAdd(&cursor[(((uint64_t)po * NB + b) * gmul + grp) * cs_out], (unsigned long long)c);
            if (at + c > cap_out) *overflow = 1ull;
#endif
        }
        gbase[b] = (uint32_t)(at < cap_out ? at : cap_out);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t idx = tid + e * PT_BLOCK;
        if (rank[e] == EMPTY) continue;
        const uint64_t at = (uint64_t)gbase[bk[e]] + rank[e];
        if (at >= cap_out) continue;                            // overflow: flagged above, the host starts over in table form
        const uint64_t o = (((uint64_t)po * NB + bk[e]) * gmul + grp) * cap_out + at;
#if MG_PT_EXP == 1
        if (tk[idx * 2] == 0x1234567ull && tcnt[idx] == 77) keys_out[o * 2] = 1;
#else
        keys_out[o * 2] = tk[idx * 2]; keys_out[o * 2 + 1] = tk[idx * 2 + 1];
        wt_out[o] = tcnt[idx];
#endif
    }
}
__global__ void __launch_bounds__(PT_FBLOCK)
k_part_fold(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ wt_in, const unsigned long long* __restrict__ n_in,
            const uint32_t* __restrict__ gslots, uint32_t gmask, uint64_t* ukeys, uint32_t U0,
            uint32_t* coo_g, uint32_t* coo_c, unsigned long long* ctr,
            uint32_t* ll_short, uint32_t* ll_long, unsigned long long* ll_n, int ll_split, uint32_t cs_in) {
    // ll_* (nullable): the new uniques' ids by length class, appended to two dense lists (ll_n = their lengths) --
    // those shorter than ll_split nt and the others.  Rounds 1 and 2 of the reference cascade take one length class each
    // (miRge.pl:326): walking "their" list they run with full lanes instead of a third / two thirds of every wave's.
    // The two list counters are bumped by two other threads at the same moment as the rank counter: issued one after
    // the other by one thread, the three round trips cost the kernel 0.19 ms (16 K partitions, each waiting for them).
    __shared__ uint32_t s_lcnt[2], s_lbase[2];
    __shared__ uint64_t tk[PT_C2 * 2];
    __shared__ uint32_t towner[2 * PT_C2];
    __shared__ uint32_t tcnt[PT_C2];
    __shared__ uint32_t s_scan[PT_FBLOCK / 64];
    __shared__ uint32_t s_base[2];
    constexpr int PER = PT_C2 / PT_FBLOCK;
    const uint32_t tid = threadIdx.x, q = blockIdx.x;
    const unsigned long long nq = n_in[(uint64_t)q * cs_in];
    const uint32_t nvalid = (uint32_t)(nq < (unsigned long long)PT_C2 ? nq : (unsigned long long)PT_C2);
    if (nvalid == 0) return;
    const uint64_t* kin = keys_in + (uint64_t)q * PT_C2 * 2;
    const uint32_t* win = wt_in + (uint64_t)q * PT_C2;
    uint32_t wcur[PER];                                          // the records' counts, fetched with the keys (as in k_part_pass)
#pragma unroll
    for (int e = 0; e < PER; ++e) { const uint32_t idx = tid + e * PT_FBLOCK; wcur[e] = idx < nvalid ? win[idx] : 0u; }
    for (uint32_t j = tid; j < nvalid * 2; j += PT_FBLOCK) tk[j] = kin[j];
    for (uint32_t j = tid; j < 2u * PT_C2; j += PT_FBLOCK) towner[j] = EMPTY;
    for (uint32_t j = tid; j < (uint32_t)PT_C2; j += PT_FBLOCK) tcnt[j] = 0;
    if (tid < 2) s_lcnt[tid] = 0;
    __syncthreads();
    uint32_t trep[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t idx = tid + e * PT_FBLOCK;
        trep[e] = EMPTY;
        if (idx < nvalid) {
            uint64_t k[2] = {tk[idx * 2], tk[idx * 2 + 1]};
            uint32_t h = key_hash<2>(k) & (2u * PT_C2 - 1);
            for (;;) {
                const uint32_t o = atomicCAS(&towner[h], EMPTY, idx);
                if (o == EMPTY) { trep[e] = idx; break; }
                if (tk[o * 2] == k[0] && tk[o * 2 + 1] == k[1]) { trep[e] = o; break; }
                h = (h + 1) & (2u * PT_C2 - 1);
            }
            atomicAdd(&tcnt[trep[e]], wcur[e]);
        }
    }
    __syncthreads();
    // representatives -> (unique id, count) entries; ids of sequences an earlier sample brought come from the
    // cross-sample table, the others are appended (as k_collapse_emit)
    uint32_t g[PER], lrank[PER];
    uint32_t nrep = 0, nnew = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const uint32_t idx = tid + e * PT_FBLOCK;
        g[e] = EMPTY; lrank[e] = 0;
        const bool rep = idx < nvalid && trep[e] == idx;
        if (rep && gslots) {
            const uint64_t k[2] = {tk[idx * 2], tk[idx * 2 + 1]};
            uint32_t h = slot_hash<2>(k) & gmask;
            for (;;) {
                const uint32_t o = gslots[h];
                if (o == EMPTY) break;
                if (ukeys[(uint64_t)o * 2] == k[0] && ukeys[(uint64_t)o * 2 + 1] == k[1]) { g[e] = o; break; }
                h = (h + 1) & gmask;
            }
        }
        nrep += rep; nnew += rep && g[e] == EMPTY;
        if (ll_short) {
            // ranks inside the block's two lists: one LDS atomic per wave and list (2048 threads on two words serialise)
            const bool isnew = rep && g[e] == EMPTY;
            const bool shrt = isnew && (int)(tk[(idx < nvalid ? idx : 0) * 2 + 1] >> 56) < ll_split;
            const unsigned long long ms = __ballot(shrt), ml = __ballot(isnew && !shrt);
            const unsigned long long below = (1ull << (tid & 63)) - 1;
            uint32_t bs = 0, bl = 0;
            if ((tid & 63) == 0) {
                if (ms) bs = atomicAdd(&s_lcnt[0], (uint32_t)__popcll(ms));
                if (ml) bl = atomicAdd(&s_lcnt[1], (uint32_t)__popcll(ml));
            }
            bs = __shfl(bs, 0); bl = __shfl(bl, 0);
            lrank[e] = shrt ? bs + (uint32_t)__popcll(ms & below) : bl + (uint32_t)__popcll(ml & below);
        }
        if (!rep) trep[e] = EMPTY;                              // from here on: EMPTY = not a representative
    }
    // inclusive scan of (representatives | new uniques << 16) over the block: inside each wave by shuffles, across the
    // 16 waves through their totals in LDS -- two barriers (a 10-step scan in LDS cost twenty)
    uint32_t incl = nrep | (nnew << 16);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if ((int)(tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < PT_FBLOCK / 64; ++w) before += w < (int)(tid >> 6) ? s_scan[w] : 0u;
        incl += before;
    }
    if (tid == PT_FBLOCK - 1) {
        const uint32_t trp = incl & 0xFFFF, tnw = incl >> 16;
#if MG_PT_EXP == 5
        const unsigned long long old = ((unsigned long long)(q * 600u) << 32) | (q * 600u); (void)trp; (void)tnw;
#else
        const unsigned long long old = atomicAdd(&ctr[0], ((unsigned long long)tnw << 32) | trp);
#endif
        s_base[0] = (uint32_t)old; s_base[1] = (uint32_t)(old >> 32);
    }
#if MG_PT_EXP == 5
    if (ll_short && tid < 2) s_lbase[tid] = q * 300u;
#else
    if (ll_short && tid < 2) s_lbase[tid] = s_lcnt[tid] ? (uint32_t)atomicAdd(&ll_n[tid * LL_CSTRIDE], (unsigned long long)s_lcnt[tid]) : 0u;
#endif   // (s_lcnt is complete: the scan's barriers lie behind)
    __syncthreads();
    uint32_t prep = s_base[0] + (incl & 0xFFFF) - nrep, pnew = s_base[1] + (incl >> 16) - nnew;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        if (trep[e] == EMPTY) continue;
        const uint32_t idx = tid + e * PT_FBLOCK;
        uint32_t gg = g[e];
        if (gg == EMPTY) {
            gg = U0 + pnew++;
            ukeys[(uint64_t)gg * 2] = tk[idx * 2]; ukeys[(uint64_t)gg * 2 + 1] = tk[idx * 2 + 1];
            if (ll_short) {
                if ((int)(tk[idx * 2 + 1] >> 56) < ll_split) ll_short[s_lbase[0] + lrank[e]] = gg;
                else ll_long[s_lbase[1] + lrank[e]] = gg;
            }
        }
        coo_g[prep] = gg; coo_c[prep] = tcnt[idx];
        ++prep;
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
    uint32_t h = slot_hash<KW>(k) & gmask;
    for (;;) {
        if (atomicCAS(&gslots[h], EMPTY, g) == EMPTY) break;
        h = (h + 1) & gmask;
    }
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
// reverse complement of a k-nt code (first nt most significant)
MG_HD uint32_t rc_code(uint32_t x, int k) { return (uint32_t)((~rev2((uint64_t)x)) >> (64 - 2 * k)) & (uint32_t)ones(2 * k); }
// ---- block schemes: shapes and codes
MG_HD int scheme_nshapes(int m) { return m == 1 ? 4 : 6; }
// shape i of the scheme as a block mask, bit j = block j (block 0 = the first B nt).  Shape 0 is the one made of the
// leading blocks: it finds every alignment without a mismatch in the first (4 - m) * B nt.
MG_HD uint32_t scheme_shape(int m, int i) { return ((m == 1 ? 0xDBE7u : 0xCA6953u) >> (4 * i)) & 15u; }
MG_HD uint32_t shape_mirror(uint32_t mask) { return ((mask & 1) << 3) | ((mask & 2) << 1) | ((mask & 4) >> 1) | ((mask & 8) >> 3); }
// nt a shape selects
MG_HD int shape_sel(uint32_t bl, uint32_t mask) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) n += ((mask >> j) & 1) ? blk(bl, j) : 0;
    return n;
}
// code of the blocks a shape selects from a W-nt window (W = the blocks' sum; first nt most significant), in window order
MG_HD uint32_t shape_code(uint64_t win, uint32_t bl, uint32_t mask) {
    uint32_t code = 0;
    int left = blocks_W(bl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int bj = blk(bl, j);
        left -= bj;
        if ((mask >> j) & 1) code = (code << (2 * bj)) | (uint32_t)((win >> (2 * left)) & ones(2 * bj));
    }
    return code;
}
// the same for four equal blocks of B nt (the kernels' instantiation for rounds whose one scheme is of this kind: fixed
// shifts instead of per-block ones -- the general form costs the one-mismatch rounds 16 B more scratch per lane)
MG_HD uint32_t shape_code_eq(uint64_t win, int B, uint32_t mask) {
    uint32_t code = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((mask >> j) & 1) code = (code << (2 * B)) | (uint32_t)((win >> (2 * B * (3 - j))) & ones(2 * B));
    return code;
}
MG_HD bool blocks_equal(uint32_t bl) { return bl == blocks_eq(blk(bl, 0)); }
// fewest / most nt a shape of the scheme selects (its least / most specific key)
MG_HD int scheme_min_sel(uint32_t bl, int m) { int v = 99; for (int i = 0; i < (m == 1 ? 4 : 6); ++i) { const int x = shape_sel(bl, ((m == 1 ? 0xDBE7u : 0xCA6953u) >> (4 * i)) & 15u); v = x < v ? x : v; } return v; }
MG_HD int scheme_max_sel(uint32_t bl, int m) { int v = 0; for (int i = 0; i < (m == 1 ? 4 : 6); ++i) { const int x = shape_sel(bl, ((m == 1 ? 0xDBE7u : 0xCA6953u) >> (4 * i)) & 15u); v = x > v ? x : v; } return v; }
// up to 32 nt of text starting at p, first nt most significant
MG_HD uint64_t text_window(const uint64_t* __restrict__ text, uint32_t p, int k) {
    uint32_t wi = p >> 5; int bs = (p & 31) * 2;
    uint64_t t0 = text[wi], t1 = text[wi + 1];
    uint64_t v = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0;
    v &= ones(2 * k);
    return rev2(v) >> (64 - 2 * k);
}
// reverse complement of a k-nt code of up to 32 nt
MG_HD uint64_t rc_code64(uint64_t x, int k) { return ((~rev2(x)) >> (64 - 2 * k)) & ones(2 * k); }
// The two buckets of text position p in a both-strand table: bf holds (p, fwd), br holds (p, rev).
//   contiguous table (bl = 0): k-mer K -> K, rc(K).
//   shape table: a forward alignment puts the read window on the text window W as it is: key = code of the shape's
//   blocks in W.  A reverse alignment puts the reverse complement of the read window there, i.e. the read window IS
//   the reverse complement of W: key = code of the shape's blocks in rc(W).
// k = most nt of a key that index the table (the key's leading nt).
MG_HD void both_buckets(const uint64_t* __restrict__ text, uint32_t p, int k, uint32_t bl, uint32_t mask, uint32_t& bf, uint32_t& br) {
    if (bl == 0) { bf = text_kmer(text, p, k); br = rc_code(bf, k); return; }
    const int W = blocks_W(bl);
    const uint64_t win = text_window(text, p, W);
    const int sel = shape_sel(bl, mask), kk = sel < k ? sel : k;
    bf = shape_code(win, bl, mask) >> (2 * (sel - kk));
    br = shape_code(rc_code64(win, W), bl, mask) >> (2 * (sel - kk));
}
__global__ void __launch_bounds__(256)
k_kmer_hist(const uint64_t* __restrict__ text, uint32_t T, int kb, uint32_t* hist) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) atomicAdd(&hist[text_kmer(text, p, kb)], 1u);
}
__global__ void __launch_bounds__(256)
k_both_hist(const uint64_t* __restrict__ text, uint32_t T, int k, uint32_t B, uint32_t mask, uint32_t* hist) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) { uint32_t bf, br; both_buckets(text, p, k, B, mask, bf, br); atomicAdd(&hist[bf], 1u); atomicAdd(&hist[br], 1u); }
}
MG_HD KEnt make_kent(const uint64_t* __restrict__ text, uint32_t p) {
    // 96 bits of text starting at nt p - CTX_BACK (zeros for nt before the text starts)
    int64_t s0 = (int64_t)p - CTX_BACK;
    uint32_t st = s0 < 0 ? 0u : (uint32_t)s0;
    uint32_t wi = st >> 5; int bs = (int)(st & 31) * 2;
    uint64_t t0 = text[wi], t1 = text[wi + 1], t2 = text[wi + 2];
    uint64_t lo = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0;
    uint64_t hi = bs ? (t1 >> bs) | (t2 << (64 - bs)) : t1;
    // Branch-free on purpose.  Written as `if (s0 < 0) { ...shifts... }` hipcc (ROCm 7.2) emits an EXEC-masked
    // block of 64-bit shifts without a skip branch, and on gfx950 that build returned wrong, run-to-run varying
    // context words for ~8 % of positions (tools/repro/exec_masked_shift_repro.hip reproduces it); this form is exact.
    const int sh = s0 < 0 ? (int)(-s0) * 2 : 0;        // 0..32
    const uint64_t carry = sh ? lo >> (64 - sh) : 0ull;
    hi = (hi << sh) | carry;
    lo = lo << sh;
    KEnt e; e.pos = p; e.c0 = (uint32_t)lo; e.c1 = (uint32_t)(lo >> 32); e.c2 = (uint32_t)hi;
    return e;
}
// pre-filter word of an entry: context nt 4 .. 35 (text[pos-12, pos+20)), i.e. bits 8 .. 71 of the 96-bit context
constexpr int PRE_SKIP = 4;
MG_HD uint64_t make_kpre(const KEnt& e) { return ((((uint64_t)e.c0 | ((uint64_t)e.c1 << 32)) >> (2 * PRE_SKIP)) | ((uint64_t)e.c2 << (64 - 2 * PRE_SKIP))); }
__global__ void __launch_bounds__(256)
k_kmer_scatter(const uint64_t* __restrict__ text, uint32_t T, int kb, uint32_t* cursor, KEnt* kent, uint64_t* kpre) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) {
        const KEnt e = make_kent(text, p);
        const uint32_t at = atomicAdd(&cursor[text_kmer(text, p, kb)], 1u);
        kent[at] = e; kpre[at] = make_kpre(e);
    }
}
// 64 bits of 2-bit text starting at nt position x (may be negative: nt before the text read as zeros)
MG_HD uint64_t text_bits64(const uint64_t* __restrict__ text, int64_t x) {
    // branch-free on purpose (see make_kent)
    const uint64_t st = x < 0 ? 0ull : (uint64_t)x;
    const int sh = x < 0 ? (x < -32 ? 64 : (int)(-x) * 2) : 0;
    const uint64_t wi = st >> 5; const int bs = (int)(st & 31) * 2;
    const uint64_t t0 = text[wi], t1 = text[wi + 1];
    const uint64_t v = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0;
    return sh >= 64 ? 0ull : v << sh;
}
// 64 bits of the N / boundary bitmap for positions x .. x + 63 (positions before the text count as set)
MG_HD uint64_t bad_bits64(const uint64_t* __restrict__ bad, int64_t x) {
    const uint64_t st = x < 0 ? 0ull : (uint64_t)x;
    const int sh = x < 0 ? (x < -64 ? 64 : (int)(-x)) : 0;
    const uint64_t wi = st >> 6; const int bs = (int)(st & 63);
    const uint64_t b0 = bad[wi], b1 = bad[wi + 1];
    const uint64_t v = bs ? (b0 >> bs) | (b1 << (64 - bs)) : b0;
    return sh >= 64 ? ~0ull : (sh ? (v << sh) | ones(sh) : v);
}
MG_HD uint32_t ref_of(const LibDev& L, uint32_t pos);
// K = length of the window the table looks up (k-mer, or 4B of a scheme)
MG_HD PEnt make_pent(const LibDev& L, uint32_t p, int K, bool rev) {
    PEnt e;
    e.pos = p | (rev ? POS_REV : 0u);
    const uint64_t right = bad_bits64(L.bad, (int64_t)p), left = bad_bits64(L.bad, (int64_t)p - 64);
    int dr = right ? __builtin_ctzll(right) : 64, dl = left ? __builtin_clzll(left) : 64;       // clean nt from p on / just before p
    if (rev) {
        // mirrored: "left" of the lookup position is what lies beyond the window's end in the text, and the other way round
        const int ml = dr - K, mr = dl + K;
        dl = ml < 0 ? 0 : ml; dr = ml < 0 ? 0 : mr;            // ml < 0: the looked-up window itself is not clean
    }
    dr = dr > 63 ? 63 : dr; dl = dl > 63 ? 63 : dl;
    e.info = ref_of(L, p) | ((uint32_t)dl << 20) | ((uint32_t)dr << 26);
    if (!rev) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint64_t v = text_bits64(L.text, (int64_t)p - PCTX_BACK + 32 * j);
            e.c[2 * j] = (uint32_t)v; e.c[2 * j + 1] = (uint32_t)(v >> 32);
        }
    } else {
        // reverse complement of text[cs, cs + 96), cs = p + K - 56: word j of the result = ~rev2(word 2 - j)
        const int64_t cs = (int64_t)p + K - (PCTX_NT - PCTX_BACK);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint64_t v = ~rev2(text_bits64(L.text, cs + 32 * (2 - j)));
            e.c[2 * j] = (uint32_t)v; e.c[2 * j + 1] = (uint32_t)(v >> 32);
        }
    }
    return e;
}
__global__ void __launch_bounds__(256)
k_both_scatter(LibDev L, int k, uint32_t B, uint32_t mask, uint32_t* cursor, PEnt* pent) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < L.T) {
        const int K = B ? blocks_W(B) : k;
        uint32_t bf, br; both_buckets(L.text, p, k, B, mask, bf, br);
        uint32_t at = atomicAdd(&cursor[bf], 1u);
        pent[at] = make_pent(L, p, K, false);
        at = atomicAdd(&cursor[br], 1u);
        pent[at] = make_pent(L, p, K, true);
    }
}
// 8-byte pre-filter word of an entry: context nt [PCTX_BACK, PCTX_BACK + 32), i.e. what a read lines up against from
// the nt its lookup started at.  A bucket list is first walked through these (a quarter of the entries' bytes): a
// candidate whose mismatches inside the word already exceed the round's allowance never has its entry fetched.
MG_HD uint64_t pent_pre(const PEnt& e) {
    const uint64_t c1 = (uint64_t)e.c[2] | ((uint64_t)e.c[3] << 32), c2 = (uint64_t)e.c[4] | ((uint64_t)e.c[5] << 32);
    return (c1 >> (2 * PCTX_BACK - 64)) | (c2 << (128 - 2 * PCTX_BACK));
}
__global__ void __launch_bounds__(256)
k_pent_pre(const PEnt* __restrict__ pent, uint64_t n, uint64_t* __restrict__ ppre) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ppre[i] = pent_pre(pent[i]);
}
// slot table of one both-strand index from its CSR form (LibDev::slots)
__global__ void __launch_bounds__(256)
k_slot_fill(const uint32_t* __restrict__ off, const PEnt* __restrict__ pent, uint64_t nb, PEnt* __restrict__ slot) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t lo = off[b], hi = off[b + 1];
    PEnt e;
    if (hi == lo + 1) e = pent[lo];
    else {
#pragma unroll
        for (int j = 0; j < 6; ++j) e.c[j] = 0;
        if (hi == lo) { e.pos = SLOT_EMPTY; e.info = 0; }
        else { e.pos = SLOT_LIST; e.info = lo; e.c[0] = hi; }
    }
    slot[b] = e;
}

// the same two steps for a compact table: entries are text position | strand, a slot is the bucket's range or its only entry
__global__ void __launch_bounds__(256)
k_both_scatter_pos(const uint64_t* __restrict__ text, uint32_t T, int k, uint32_t B, uint32_t mask, uint32_t* cursor, uint32_t* __restrict__ cpos) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < T) {
        uint32_t bf, br; both_buckets(text, p, k, B, mask, bf, br);
        cpos[atomicAdd(&cursor[bf], 1u)] = p;
        cpos[atomicAdd(&cursor[br], 1u)] = p | POS_REV;
    }
}
__global__ void __launch_bounds__(256)
k_cslot_fill(const uint32_t* __restrict__ off, const uint32_t* __restrict__ cpos, uint64_t nb, CSlot* __restrict__ slot) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t lo = off[b], hi = off[b + 1];
    CSlot e{lo, hi};
    if (hi == lo + 1) { e.a = cpos[lo]; e.b = CS_SINGLE; }
    slot[b] = e;
}

// bucket occupancy bitmap from the histogram: most exact-length lookups of reads that do not belong to
// a library end in an empty bucket, and this answers them from a structure 32x smaller than koff
__global__ void __launch_bounds__(256)
k_bucket_bits(const uint32_t* __restrict__ hist, uint64_t nb, uint32_t* kbits) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 32 >= nb) return;
    uint32_t bits = 0;
    for (int j = 0; j < 32; ++j) { uint64_t b = w * 32 + j; if (b < nb && hist[b]) bits |= 1u << j; }
    kbits[w] = bits;
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

// no reference N and no reference boundary inside [w, w+l) ?
template <int SW>
MG_HD bool window_clean(const LibDev& L, int64_t w, int l) {
    uint32_t bi = (uint32_t)(w >> 6); int sh = (int)(w & 63);
    uint64_t b0 = L.bad[bi], b1 = L.bad[bi + 1];
    uint64_t r0 = sh ? (b0 >> sh) | (b1 << (64 - sh)) : b0;
    uint64_t any = r0 & ones(l);
    if (SW > 2) {
        uint64_t b2 = L.bad[bi + 2];
        uint64_t r1 = sh ? (b1 >> sh) | (b2 << (64 - sh)) : b1;
        any |= r1 & ones(l - 64);
    }
    return any == 0;
}

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
    if (!window_clean<SW>(L, w, q.l)) return false;
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

// largest j with ref_start[j] <= pos.  A binary search over all references is ~15 DEPENDENT loads for a 20k-40k
// entry library, three times per aligned read: the segment table narrows it to the references that begin inside
// one 2^lut_shift-nt segment (one 8-byte load, then 0-2 steps).
MG_HD uint32_t ref_of(const LibDev& L, uint32_t pos) {
    const uint32_t sgm = pos >> L.lut_shift;
    uint32_t lo = L.ref_lut[sgm], hi = L.ref_lut[sgm + 1] + 1;
    if (hi > L.n_refs) hi = L.n_refs;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (L.ref_start[mid] <= pos) lo = mid; else hi = mid; }
    return lo;
}
// host: the segment table of ref_of (n + 1 starts, text length T)
inline int ref_lut_shift(uint64_t T) { int sh = 5; while ((T >> sh) > (1u << 20)) ++sh; return sh; }
template <class Vec> inline void build_ref_lut(const uint32_t* ref_start, uint32_t n, uint64_t T, int shift, Vec& lut) {
    const uint64_t nseg = (T >> shift) + 2;
    lut.assign((size_t)nseg, 0);
    uint32_t j = 0;
    for (uint64_t sg = 0; sg < nseg; ++sg) {
        const uint64_t pos = sg << shift;
        while (j + 1 < n && ref_start[j + 1] <= pos) ++j;
        lut[(size_t)sg] = j;
    }
}

// Pigeonhole pieces.  An alignment has <= seedmm mismatches in the seed and <= maxmm overall, so one
// of np = min(seedmm, maxmm) + 1 pieces of the seed matches exactly; when that equals maxmm + 1 the
// pieces are cut from the whole read instead (longer pieces, same guarantee).  np is the same for
// every read of a round, so the piece loop is wave-uniform.
MG_HD int round_np(const RoundDev& P) { return (P.seedmm < P.maxmm ? P.seedmm : P.maxmm) + 1; }
MG_HD bool round_whole(const RoundDev& P) { return round_np(P) == P.maxmm + 1; }
// Pigeonhole, the other way round: once pieces 0 .. p-1 of the SEED have been looked up, every alignment with fewer
// than p seed mismatches is among their candidates (one of those p pieces is free of mismatches).  So if the best
// alignment known by then already has < p seed mismatches, piece p can only turn up worse strata, which
// `--best --strata` drops: its lookups are skipped.  (Not for pieces cut from the whole read: there the count that
// is bounded is the total, not the stratum.)  `score` = best hit_score so far, 0 = none.
MG_HD bool piece_needed(const RoundDev& P, int p, uint64_t score) {
    if (p == 0 || score == 0 || round_whole(P)) return true;
    return 4 - (int)(score >> 36) >= p;
}
// floor(x * p / np) for x <= 4095, p <= np <= 4, without an integer division
MG_HD int piece_cut(int x, int p, int np) { return (int)(((uint32_t)(x * p) * (uint32_t)(65536 / np + 1)) >> 16); }

// A read prepared for one round: both orientations, seed length, span the pieces are cut from.
template <int SW> struct Query {
    Oriented<SW> q[2];                          // [1] forward, [0] reverse complement
    int l, s, span;
    bool ok;                                    // false: bowtie skips the read (empty / l <= seedmm)
};
template <int KW>
MG_HD void query_setup(const uint64_t* key, const RoundDev& P, Query<KeyTraits<KW>::SW>& Q) {
    constexpr int SW = KeyTraits<KW>::SW;
    key_to_read<KW>(key, P.trim5, P.trim3, Q.q[1]);
    const int l = Q.q[1].l;
    Q.l = l; Q.ok = !(l <= 0 || l <= P.seedmm);
    Q.s = 0; Q.span = 0;
    if (!Q.ok) { Q.q[0] = Q.q[1]; return; }
    revcomp<SW>(Q.q[1], Q.q[0]);
    Q.s = P.seedlen < l ? P.seedlen : l;
    Q.span = round_whole(P) ? l : Q.s;
}
// the same without the reverse complement (q[0] is left unset): bucket codes of both strands come from the forward
// read (piece_buckets_fwd), so the kernel builds the reverse complement only for a piece that has candidates
template <int KW>
MG_HD void query_setup_fwd(const uint64_t* key, const RoundDev& P, Query<KeyTraits<KW>::SW>& Q) {
    key_to_read<KW>(key, P.trim5, P.trim3, Q.q[1]);
    const int l = Q.q[1].l;
    Q.l = l; Q.ok = !(l <= 0 || l <= P.seedmm);
    Q.s = 0; Q.span = 0;
    if (!Q.ok) return;
    Q.s = P.seedlen < l ? P.seedlen : l;
    Q.span = round_whole(P) ? l : Q.s;
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
// piece [ps, pe) (read coordinates) on strand fw -> where it sits in the oriented read (qs) and
// its bucket range [b0, b1] in koff
template <int SW, bool HASN>
MG_HD bool piece_buckets(const Oriented<SW>& o, int l, int kb, int fw, int ps, int pe, int& qs, uint64_t& b0, uint64_t& b1) {
    qs = fw ? ps : l - pe;
    int kq = pe - ps < kb ? pe - ps : kb;
    if (kq <= 0) return false;
    if (HASN && n_in_range<SW>(o, qs, kq)) return false;      // a piece holding a read N is not the exact one
    uint32_t code = kmer_code<SW>(o, qs, kq);
    int sh = 2 * (kb - kq);
    b0 = (uint64_t)code << sh; b1 = ((uint64_t)code + 1) << sh;
    return true;
}

// The same from the FORWARD read for either strand.  The reverse-complement piece starts at l - pe of the reversed
// read and its first kq nt are the complements of forward nt pe-1, pe-2, ..., pe-kq: read least-significant-first
// from the forward bits that is already the most-significant-first code, complemented (A=0 C=1 G=2 T=3: comp = ~).
template <int SW, bool HASN>
MG_HD bool piece_buckets_fwd(const Oriented<SW>& fwd, int l, int kb, int fw, int ps, int pe, int& qs, uint64_t& b0, uint64_t& b1) {
    if (fw) return piece_buckets<SW, HASN>(fwd, l, kb, 1, ps, pe, qs, b0, b1);
    qs = l - pe;
    const int kq = pe - ps < kb ? pe - ps : kb;
    if (kq <= 0) return false;
    if (HASN && n_in_range<SW>(fwd, pe - kq, kq)) return false;
    const uint32_t code = (uint32_t)(~bits_at<SW>(fwd.s, 2 * (pe - kq)) & ones(2 * kq));
    const int sh = 2 * (kb - kq);
    b0 = (uint64_t)code << sh; b1 = ((uint64_t)code + 1) << sh;
    return true;
}

// The read of one (orientation, piece), laid out in the coordinates of an index entry's context:
// read nt i sits at context nt i - a + j0 (a = first read nt the context can cover, j0 = context nt
// it lines up with).  ra = read bits, m = nts to compare (covered, not a read N), sd = the seed nts
// among them.  full: the context covers the whole window, so the entry alone decides.
struct PieceAl { uint64_t ra0, m0, sd0; uint32_t ra1, m1, sd1; uint32_t lo, hi; uint32_t meta; };
//   meta: owner (bits 0-7) | fw (8) | full (9) | qs (10-16) | l (17-23) | nN (24-26) | nNseed (27-29)
MG_HD uint32_t pa_meta(uint32_t owner, int fw, bool full, int qs, int l, int nN, int nNseed) {
    return owner | ((uint32_t)fw << 8) | ((uint32_t)full << 9) | ((uint32_t)qs << 10) | ((uint32_t)l << 17) |
           ((uint32_t)nN << 24) | ((uint32_t)nNseed << 27);
}
// 96 bits of the (SW+1)-word value [0, x] starting at bit `bp` of it (i.e. x shifted left by 64 - bp)
template <int SW> MG_HD void take96(const uint64_t (&x)[SW], int bp, uint64_t& w0, uint32_t& w1) {
    uint64_t y[SW + 1];
    y[0] = 0;
#pragma unroll
    for (int i = 0; i < SW; ++i) y[i + 1] = x[i];
    w0 = bits_at<SW + 1>(y, bp);
    w1 = (uint32_t)bits_at<SW + 1>(y, bp + 64);
}
template <int SW, bool HASN>
MG_HD void piece_align(const Oriented<SW>& o, const Strand<SW, HASN>& S, int qs, int fw, uint32_t owner, PieceAl& A) {
    const int a = qs > CTX_BACK ? qs - CTX_BACK : 0;
    const int j0 = qs < CTX_BACK ? CTX_BACK - qs : 0;
    int nov = o.l - a; if (nov > CTX_NT - j0) nov = CTX_NT - j0;
    const int bp = 64 + 2 * a - 2 * j0;                       // >= 32
    take96<SW>(o.s, bp, A.ra0, A.ra1);
    const int jlo = j0, jhi = j0 + nov;
    A.m0 = (ones(2 * jhi) ^ ones(2 * jlo)) & M55;
    A.m1 = (uint32_t)(ones(2 * jhi - 64) & M55);
    uint64_t s0; uint32_t s1;
    take96<SW>(S.seedm, bp, s0, s1);
    A.sd0 = s0 & A.m0; A.sd1 = s1 & A.m1;
    if (HASN) {
        uint64_t n0; uint32_t n1;
        take96<SW>(S.nsp, bp, n0, n1);
        A.m0 &= ~n0; A.m1 &= ~n1; A.sd0 &= ~n0; A.sd1 &= ~n1;
    }
    A.meta = pa_meta(owner, fw, (a == 0) && (nov == o.l), qs, o.l, S.nN, S.nNseed);
}
// One index entry against a prepared piece: 0 = reject, 1 = valid (sm, tm, w set),
// 2 = passes the context pre-filter but the context does not cover the window: finish on the text.
template <int SW>
MG_HD int cand_check(const LibDev& L, const PieceAl& A, const KEnt& e, const RoundDev& P, int& sm, int& tm, uint32_t& w) {
    const uint32_t qs = (A.meta >> 10) & 127, l = (A.meta >> 17) & 127;
    if (e.pos < qs) return 0;
    w = e.pos - qs;
    if (w + l > L.T) return 0;
    const uint64_t c64 = (uint64_t)e.c0 | ((uint64_t)e.c1 << 32);
    uint64_t d0 = c64 ^ A.ra0; d0 = (d0 | (d0 >> 1)) & A.m0;
    uint32_t d1 = e.c2 ^ A.ra1; d1 = (d1 | (d1 >> 1)) & A.m1;
    tm = (int)((A.meta >> 24) & 7) + popc64(d0) + popc64((uint64_t)d1);
    if (tm > P.maxmm) return 0;
    if (!((A.meta >> 9) & 1)) return 2;
    sm = (int)((A.meta >> 27) & 7) + popc64(d0 & A.sd0) + popc64((uint64_t)(d1 & A.sd1));
    if (sm > P.seedmm) return 0;
    return window_clean<SW>(L, (int64_t)w, (int)l) ? 1 : 0;
}

// Long candidate lists (range queries of short pieces: ~55 entries per lookup for a 16-nt read in the 1-mismatch
// rounds, ~280 for a 13-nt isomiR query) are bandwidth-bound on the 16-byte entries, and nearly every candidate
// already fails on the 32 nt around its position.  They are therefore scanned through kpre (8 bytes per entry);
// only survivors load the full entry.  PreAl = the prepared piece cut down to the pre-filter window.
struct PreAl { uint64_t ra, m; int nN; };
MG_HD PreAl pre_of(const PieceAl& A) {
    PreAl R;
    R.ra = (A.ra0 >> (2 * PRE_SKIP)) | ((uint64_t)A.ra1 << (64 - 2 * PRE_SKIP));
    R.m = (A.m0 >> (2 * PRE_SKIP)) | ((uint64_t)A.m1 << (64 - 2 * PRE_SKIP));
    R.nN = (int)((A.meta >> 24) & 7);
    return R;
}
MG_HD bool pre_pass(const PreAl& R, uint64_t pv, int maxmm) {
    uint64_t d = pv ^ R.ra; d = (d | (d >> 1)) & R.m;
    return R.nN + popc64(d) <= maxmm;
}

// both ends of a bucket range with one 8-byte load when they are neighbours
MG_HD void koff_range(const LibDev& L, uint64_t b0, uint64_t b1, uint32_t& lo, uint32_t& hi) {
    lo = 0; hi = 0;
    if (b1 == b0 + 1) {
        // the occupancy bit first -- unless koff itself is small enough to stay cache-resident (kbits == null): then
        // asking it would only put one more load in front of the dependent chain bitmap -> offsets -> entry
        if (L.kbits && !((L.kbits[b0 >> 5] >> (b0 & 31)) & 1)) return;          // empty bucket
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
        u32x2 v = *reinterpret_cast<const u32x2*>(L.koff + b0);
        lo = v.x; hi = v.y;
    } else {
        lo = L.koff[b0]; hi = L.koff[b1];
    }
}

// ==========================================================================================
// The probe path: rounds against a library that carries the both-strand slot tables (LibDev::slots), 16-byte keys,
// at most three seed pieces.  A read makes one lookup per piece -- occupancy bit, then the slot: ONE random line
// that answers both strands and usually holds the candidate itself, self-contained (PEnt) -- and the candidate is
// judged by shifting the entry's context into the read's coordinates (no per-piece layout of the read as in the
// generic matcher).  Lookups do not wait for each other: a wave's dependent memory steps are  key -> bits -> slots
// for two lookups at a time, instead of  bits -> offsets -> entries -> text  per piece and strand in turn.  A read
// whose pieces are shorter than the index k-mer goes through a block scheme if the library carries one that fits.
// Reads the path cannot serve (no fitting scheme, a bucket with more than PROBE_MAX_LIST entries, a full queue) are
// handed to the generic kernel.
constexpr int PROBE_MAX_LIST = 4096;

// does the round / library pair run on the probe path (host-side dispatch; 16-byte key class only)
MG_HD bool probe_round(const LibDev& L, const RoundDev& P) { return L.slots != nullptr && round_np(P) <= 3; }
// The schemes a round may use: those that cover the mismatches allowed inside the seed, min(seedmm, maxmm) -- and,
// where a scheme window can reach beyond a short seed, those that cover the whole-read allowance maxmm -- most
// specific first, at most ROUND_SCHEMES.
MG_HD int scheme_rank(const TabDesc& t) { return scheme_min_sel(tab_blocks(t), tab_m(t)) * 64 + tab_W(t); }      // by its least specific key, then by its window
MG_HD RoundSch round_schemes(const LibDev& L, const RoundDev& P) {
    RoundSch RS; RS.n = 0;
    const int ms = P.seedmm < P.maxmm ? P.seedmm : P.maxmm;
    for (int j = 0; j < ROUND_SCHEMES; ++j) { RS.t[j].slot_off = 0; RS.t[j].bits_off = 0; RS.t[j].info = 0; }
    for (int j = 0; j < L.nsch; ++j) {
        const TabDesc t = L.sch[j];
        const int m = tab_m(t);
        if (!(m == ms || (m == P.maxmm && P.seedlen < 20))) continue;
        int at = RS.n;
        while (at > 0 && scheme_rank(RS.t[at - 1]) < scheme_rank(t)) --at;
        if (at >= ROUND_SCHEMES) continue;
        for (int q = (RS.n < ROUND_SCHEMES ? RS.n : ROUND_SCHEMES - 1); q > at; --q) RS.t[q] = RS.t[q - 1];
        RS.t[at] = t;
        if (RS.n < ROUND_SCHEMES) ++RS.n;
    }
    return RS;
}
constexpr int PROBE_MAX_LOOK = 6;
// A read prepared for the probe path.
struct ProbeRead {
    uint64_t r[2];            // the read after -5 / -3, 2-bit, nt 0 in the low bits
    uint64_t ml[2], ms[2];    // masks (one bit per nt, at even positions) of its l nt / of its s seed nt
    int l, s;                 // length, seed length
    TabDesc tab;              // the table(s) its lookups go to: the contiguous one (B = 0) or a scheme's
    int nlook;                // lookups: pieces, or the scheme's shapes
    int ps[3];                // plain pieces: where their k-mers start in the read
    bool lead;                // scheme: shape 0 lies inside the seed, so it finds every stratum-0 alignment
    bool ok, served;          // ok: bowtie aligns the read at all; served: the probe path can decide it
    uint64_t nraw;            // 48-byte class: the read's N, one bit per nt (a read N mismatches everything; 0 for 16-byte keys)
};
MG_HD void probe_masks(ProbeRead& R) {
    R.ml[0] = ones(2 * R.l) & M55; R.ml[1] = ones(2 * R.l - 64) & M55;
    R.ms[0] = ones(2 * R.s) & M55; R.ms[1] = ones(2 * R.s - 64) & M55;
}
// The read from nt ps on (what a pre-filter word is compared with) and the mask of the nt that word covers.
MG_HD void pre_query(const uint64_t r0, const uint64_t r1, int l, int ps, uint64_t& rr, uint64_t& mask) {
    const int sh = 2 * ps;
    rr = sh == 0 ? r0 : (sh < 64 ? (r0 >> sh) | (r1 << (64 - sh)) : r1 >> (sh - 64));
    const int n = l - ps < 32 ? l - ps : 32;
    mask = ones(2 * n) & M55;
}
// lower bound of the candidate's mismatches: those among the nt the word covers
MG_HD int pre_mism(uint64_t rr, uint64_t mask, uint64_t pre) {
    const uint64_t d = rr ^ pre;
    return popc64((d | (d >> 1)) & mask);
}
// NS = schemes of RS looked at (a template parameter of the kernel: rounds without schemes carry none of it)
// KW = 6: a read of the 48-byte class -- served if it is at most 56 nt after -5 / -3 (two words, as a 16-byte key's);
// its N count as mismatches in probe_verify, and a lookup whose key holds one is left out (probe_lookup_n)
template <int NS, int KW = 2>
MG_HD void probe_setup(const uint64_t* key, const LibDev& L, const RoundDev& P, const RoundSch& RS, ProbeRead& R) {
    Query<KeyTraits<KW>::SW> Q;
    query_setup_fwd<KW>(key, P, Q);
    R.r[0] = Q.q[1].s[0]; R.r[1] = Q.q[1].s[1];
    R.l = Q.ok ? Q.l : 0; R.s = Q.s; R.ok = Q.ok;
    R.nraw = KeyTraits<KW>::HASN ? Q.q[1].n[0] : 0ull;
    R.tab = L.ctab; R.nlook = 0; R.lead = false; R.served = false;
    R.ps[0] = 0; R.ps[1] = 0; R.ps[2] = 0;
    if (KW == 6 && Q.ok && Q.l > 56) { R.l = 0; R.s = 0; probe_masks(R); return; }       // not served: the generic kernel's
    probe_masks(R);
    if (!Q.ok) return;
    if (KW == 6 && popc64(R.nraw) > P.maxmm) { R.served = true; return; }                // served: no alignment can exist
    const int np = round_np(P);
    if (piece_cut(Q.span, 1, np) >= L.kb) {                        // the shortest piece is the first: floor(span / np)
        R.nlook = np; R.served = true;
#pragma unroll
        for (int p = 1; p < 3; ++p) R.ps[p] = p < np ? piece_cut(Q.span, p, np) : 0;
        return;
    }
    // a block scheme: its 4B-nt window must lie inside the read, and cover the mismatches allowed there -- inside
    // the seed min(seedmm, maxmm), else maxmm.  RS is sorted: the first that fits is the most specific.
    const int ms = P.seedmm < P.maxmm ? P.seedmm : P.maxmm;
    bool found = false;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int W = tab_W(RS.t[j]), m = tab_m(RS.t[j]);
        const bool fits = !found && j < RS.n && W <= Q.l && m >= (W <= Q.s ? ms : P.maxmm);
        if (fits) { R.tab = RS.t[j]; found = true; }
    }
    if (!found) return;
    R.nlook = scheme_nshapes(tab_m(R.tab)); R.served = true;
    R.lead = shape_sel(tab_blocks(R.tab), scheme_shape(tab_m(R.tab), 0)) <= Q.s;
}
// lookup i (< R.nlook): slot index, occupancy bit index (used if tab_bits(R.tab)), and where the looked-up window
// starts in the read (ps; a scheme's window starts the read)
template <bool EQ = false>
MG_HD void probe_lookup(const ProbeRead& R, const LibDev& L, int i, uint32_t& slot, uint32_t& b, int& ps) {
    const uint32_t bl = tab_blocks(R.tab); const int kbs = tab_kbs(R.tab);
    if (bl == 0) {
        Oriented<2> o; o.s[0] = R.r[0]; o.s[1] = R.r[1]; o.n[0] = 0; o.l = R.l;
        ps = i == 0 ? R.ps[0] : (i == 1 ? R.ps[1] : R.ps[2]);
        b = kmer_code<2>(o, ps, L.kb);
    } else {
        const int m = tab_m(R.tab);
        const uint32_t mask = scheme_shape(m, i);
        if (EQ) {
            const int B = blk(bl, 0), sel = (4 - m) * B;
            const uint64_t win = rev2(R.r[0] & ones(8 * B)) >> (64 - 8 * B);
            b = shape_code_eq(win, B, mask) >> (2 * (sel - kbs));          // (kbs = min(sel, kb))
        } else {
            const int W = blocks_W(bl), sel = shape_sel(bl, mask), kk = sel < kbs ? sel : kbs;
            const uint64_t win = rev2(R.r[0] & ones(2 * W)) >> (64 - 2 * W);   // first W nt of the read, first nt most significant
            b = shape_code(win, bl, mask) >> (2 * (sel - kk));
        }
        ps = 0;
        b += (uint32_t)i << (2 * kbs);                                     // shape i's table
    }
    slot = R.tab.slot_off + b;
    b += R.tab.bits_off << 5;
}
// does the key of lookup i hold a read N?  (Then that piece / those blocks are not the exact ones: the lookup is left
// out, as the generic matcher leaves out a piece that holds an N.)
MG_HD bool probe_lookup_n(const ProbeRead& R, const LibDev& L, int i) {
    const uint32_t bl = tab_blocks(R.tab);
    if (bl == 0) {
        const int ps = i == 0 ? R.ps[0] : (i == 1 ? R.ps[1] : R.ps[2]);
        return ((R.nraw >> ps) & ones(L.kb)) != 0;
    }
    const uint32_t mask = scheme_shape(tab_m(R.tab), i);
    uint64_t km = 0; int at = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int bj = blk(bl, j); if ((mask >> j) & 1) km |= ones(bj) << at; at += bj; }
    return (R.nraw & km) != 0;
}
// may lookup i be left out, given the best alignment found so far (0 = none)?  Scheme: shape 0 has already found
// every stratum-0 alignment.  Plain pieces: piece_needed's rule.
MG_HD bool probe_needed(const ProbeRead& R, const RoundDev& P, int i, uint64_t score) {
    if (i == 0 || score == 0) return true;
    if (tab_blocks(R.tab)) return !(R.lead && (score >> 36) == 4);
    return piece_needed(P, i, score);
}
// One entry against a read whose lookup started ps nt into it: false = no alignment, true = valid alignment (f, sm,
// tm, w, ref set).  K = the looked-up window's length.  Forward and (mirrored) reverse entries are judged alike.
template <bool HASN = false>
MG_HD bool probe_verify(const RoundDev& P, const ProbeRead& R, int K, int ps, const PEnt& e,
                        int& f, int& sm, int& tm, uint32_t& w, uint32_t& ref) {
    // no N, no reference boundary inside the window: it reaches ps nt back from the lookup position and l - ps nt on
    if (ps > (int)((e.info >> 20) & 63) || R.l - ps > (int)(e.info >> 26)) return false;
    const int o = PCTX_BACK - ps;                                          // 0 .. 40: context nt the read starts at
    const uint64_t c0 = (uint64_t)e.c[0] | ((uint64_t)e.c[1] << 32), c1 = (uint64_t)e.c[2] | ((uint64_t)e.c[3] << 32),
                   c2 = (uint64_t)e.c[4] | ((uint64_t)e.c[5] << 32);
    const int sh = (2 * o) & 63;
    const uint64_t y0 = o >= 32 ? c1 : c0, y1 = o >= 32 ? c2 : c1, y2 = o >= 32 ? 0ull : c2;
    const uint64_t x0 = sh ? (y0 >> sh) | (y1 << (64 - sh)) : y0, x1 = sh ? (y1 >> sh) | (y2 << (64 - sh)) : y1;
    uint64_t d0 = x0 ^ R.r[0], d1 = x1 ^ R.r[1];
    d0 = d0 | (d0 >> 1); d1 = d1 | (d1 >> 1);
    if (HASN) { d0 |= spread32((uint32_t)R.nraw); d1 |= spread32((uint32_t)(R.nraw >> 32)); }      // a read N mismatches everything
    d0 &= R.ml[0]; d1 &= R.ml[1];
    tm = popc64(d0) + popc64(d1);
    if (tm > P.maxmm) return false;
    sm = popc64(d0 & R.ms[0]) + popc64(d1 & R.ms[1]);
    if (sm > P.seedmm) return false;
    const uint32_t pos = e.pos & POS_MASK;
    f = (e.pos >> 31) ? 0 : 1;
    w = f ? pos - (uint32_t)ps : pos - (uint32_t)(R.l - ps - K);          // >= 0 and w + l <= T follow from dl / dr
    ref = e.info & (PENT_MAX_REFS - 1);
    return true;
}
// slot -> 0 = empty, 1 = list [lo, hi) of `pent`, 2 = the entry itself
MG_HD int slot_kind(const PEnt& e, uint32_t& lo, uint32_t& hi) {
    if (e.pos == SLOT_EMPTY) return 0;
    if (e.pos == SLOT_LIST) { lo = e.info; hi = e.c[0]; return 1; }
    return 2;
}
MG_HD PEnt load_pent(const PEnt* __restrict__ a, uint64_t i) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* q = reinterpret_cast<const u32x4*>(a + i);
    // (streaming / nontemporal loads here, meant to keep slot lines from pushing the occupancy bitmaps out of the L2, lost:
    //  round 3 0.73 -> 0.82 ms, round 5 0.61 -> 1.03 ms -- the lines then miss the Infinity Cache as well)
    const u32x4 v0 = q[0], v1 = q[1];
    PEnt e; e.pos = v0.x; e.info = v0.y; e.c[0] = v0.z; e.c[1] = v0.w; e.c[2] = v1.x; e.c[3] = v1.y; e.c[4] = v1.z; e.c[5] = v1.w;
    return e;
}
// the three order keys of an alignment (0 = none): the representative by (stratum, mismatches, position, strand);
// the largest / smallest reference inside the best stratum -- they differ iff the assignment is ambiguous
struct ProbeBest { uint64_t score, rmax, rmin; uint32_t ref; };
MG_HD void probe_record(ProbeBest& Bst, int sm, int tm, uint32_t w, int f, uint32_t ref) {
    const uint64_t x = hit_score(sm, tm, w, f), y = hit_kmax(sm, ref), z = hit_kmin(sm, ref);
    if (x > Bst.score) { Bst.score = x; Bst.ref = ref; }
    Bst.rmax = y > Bst.rmax ? y : Bst.rmax; Bst.rmin = z > Bst.rmin ? z : Bst.rmin;
}
MG_HD int tab_K(const TabDesc& t, const LibDev& L) { return tab_blocks(t) ? tab_W(t) : L.kb; }

// Serial form of the probe path for one read: false = not served (ask the generic matcher).
template <int KW>
__host__ __device__ inline bool probe_one(const uint64_t* key, const LibDev& L, const RoundDev& P, ProbeBest& Bst) {
    const RoundSch RS = round_schemes(L, P);
    ProbeRead R;
    probe_setup<ROUND_SCHEMES, KW>(key, L, P, RS, R);
    Bst.score = 0; Bst.rmax = 0; Bst.rmin = 0; Bst.ref = 0;
    if (!R.ok) return true;                      // bowtie skips the read: served, no alignment
    if (!R.served) return false;
    const int K = tab_K(R.tab, L);
    for (int i = 0; i < R.nlook; ++i) {
        if (!probe_needed(R, P, i, Bst.score)) continue;
        if (KW == 6 && probe_lookup_n(R, L, i)) continue;
        uint32_t slot, b; int ps;
        probe_lookup(R, L, i, slot, b, ps);
        if (tab_bits(R.tab) && !((L.sbits[b >> 5] >> (b & 31)) & 1)) continue;
        const PEnt e = load_pent(L.slots, slot);
        uint32_t lo = 0, hi = 0;
        const int kind = slot_kind(e, lo, hi);
        if (kind == 0) continue;
        if (kind == 1 && hi - lo > (uint32_t)PROBE_MAX_LIST) return false;
        for (uint32_t c = lo; c < (kind == 2 ? lo + 1 : hi); ++c) {
            int f, sm, tm; uint32_t w, ref;
            if (probe_verify<KW == 6>(P, R, K, ps, kind == 2 ? e : load_pent(L.pent, c), f, sm, tm, w, ref)) probe_record(Bst, sm, tm, w, f, ref);
        }
    }
    return true;
}

// ---- the LDS path: libraries with compact tables (LibDev::cslots).  Lookups as on the probe path (same setup, same
// buckets); a candidate is an entry e = text position of the looked-up window | strand and is verified against the
// TEXT -- which k_lds_round keeps in LDS -- instead of against a self-contained 32-byte entry.
MG_HD bool lds_round(const LibDev& L, const RoundDev& P) { return L.cslots != nullptr && round_np(P) <= 3; }
// reverse complement of a read of l <= 64 nt held in two words
MG_HD void revcomp2(uint64_t r0, uint64_t r1, int l, uint64_t& c0, uint64_t& c1) {
    const uint64_t t0 = ~rev2(r1), t1 = ~rev2(r0);               // the reversed complement, left-aligned in 128 bits
    const int sh = 128 - 2 * l;                                  // >= 0
    const uint64_t a = sh >= 64 ? t1 : t0, b = sh >= 64 ? 0ull : t1;
    const int s6 = sh & 63;
    c0 = s6 ? (a >> s6) | (b << (64 - s6)) : a;
    c1 = s6 ? b >> s6 : b;
    c0 &= ones(2 * l); c1 &= ones(2 * l - 64);
}
// One entry against a read whose lookup (a window of K nt) started ps nt into it.  q0/q1 = the read (forward entry) or
// its reverse complement (reverse entry: bit 31 of e); l, s = length and seed length.  true = valid alignment.
MG_HD bool pos_verify(const RoundDev& P, const uint64_t* text, const uint64_t* bad, uint32_t T, uint64_t q0, uint64_t q1,
                      int l, int s, int K, int ps, uint32_t e, int& f, int& sm, int& tm, uint32_t& w) {
    f = (e >> 31) ? 0 : 1;
    const int64_t ww = (int64_t)(e & POS_MASK) - (f ? ps : l - ps - K);
    if (ww < 0 || ww + l > (int64_t)T) return false;
    w = (uint32_t)ww;
    const uint32_t wi = w >> 5; const int bs = (int)(w & 31) * 2;
    const uint64_t t0 = text[wi], t1 = text[wi + 1], t2 = text[wi + 2];
    const uint64_t x0 = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0, x1 = bs ? (t1 >> bs) | (t2 << (64 - bs)) : t1;
    uint64_t d0 = x0 ^ q0, d1 = x1 ^ q1;
    d0 = (d0 | (d0 >> 1)) & M55 & ones(2 * l); d1 = (d1 | (d1 >> 1)) & M55 & ones(2 * l - 64);
    tm = popc64(d0) + popc64(d1);
    if (tm > P.maxmm) return false;
    // the seed = the read's first s nt: the first s of a forward alignment, the last s of the reverse complement
    const int sa = f ? 0 : l - s, sb = f ? s : l;
    sm = popc64(d0 & (ones(2 * sb) ^ ones(2 * sa))) + popc64(d1 & (ones(2 * sb - 64) ^ ones(2 * sa - 64)));
    if (sm > P.seedmm) return false;
    const uint32_t bi = w >> 6; const int sh = (int)(w & 63);
    const uint64_t b0 = bad[bi], b1 = bad[bi + 1];
    const uint64_t rb = sh ? (b0 >> sh) | (b1 << (64 - sh)) : b0;
    return (rb & ones(l)) == 0;                                  // no reference N, no reference boundary inside the window (l <= 56)
}
// the three order keys of the generic matcher (positions, not references): score, max / min position of the best stratum
struct PosBest { uint64_t score, kmax, kmin; };
MG_HD void pos_record(PosBest& Bst, int sm, int tm, uint32_t w, int f) {
    const uint64_t x = hit_score(sm, tm, w, f), y = hit_kmax(sm, w), z = hit_kmin(sm, w);
    Bst.score = x > Bst.score ? x : Bst.score; Bst.kmax = y > Bst.kmax ? y : Bst.kmax; Bst.kmin = z > Bst.kmin ? z : Bst.kmin;
}
constexpr int LDS_MAX_LIST = 4096;
// Serial form of the LDS path for one read: false = not served (ask the generic matcher).
__host__ __device__ inline bool lds_one(const uint64_t* key, const LibDev& L, const RoundDev& P, PosBest& Bst) {
    const RoundSch RS = round_schemes(L, P);
    ProbeRead R;
    probe_setup<ROUND_SCHEMES>(key, L, P, RS, R);
    Bst.score = 0; Bst.kmax = 0; Bst.kmin = 0;
    if (!R.ok) return true;
    if (!R.served) return false;
    uint64_t c0, c1;
    revcomp2(R.r[0], R.r[1], R.l, c0, c1);
    const int K = tab_K(R.tab, L);
    for (int i = 0; i < R.nlook; ++i) {
        if (!probe_needed(R, P, i, Bst.score)) continue;
        uint32_t slot, b; int ps;
        probe_lookup(R, L, i, slot, b, ps);
        if (tab_bits(R.tab) && !((L.sbits[b >> 5] >> (b & 31)) & 1)) continue;
        const CSlot sl = L.cslots[slot];
        if (sl.b != CS_SINGLE && sl.b - sl.a > (uint32_t)LDS_MAX_LIST) return false;
        const uint32_t n = sl.b == CS_SINGLE ? 1u : sl.b - sl.a;
        for (uint32_t c = 0; c < n; ++c) {
            const uint32_t e = sl.b == CS_SINGLE ? sl.a : L.cpos[sl.a + c];
            int f, sm, tm; uint32_t w;
            const bool rev = e >> 31;
            if (pos_verify(P, L.text, L.bad, L.T, rev ? c0 : R.r[0], rev ? c1 : R.r[1], R.l, R.s, K, ps, e, f, sm, tm, w)) pos_record(Bst, sm, tm, w, f);
        }
    }
    return true;
}

// Serial form of the matcher (one read, all candidates in turn): the definition the kernel below
// parallelises, and what tests/hostsim runs on the CPU.
template <int KW>
__host__ __device__ inline void match_generic(const uint64_t* key, const LibDev& L, const RoundDev& P, uint64_t& score, uint64_t& kmax, uint64_t& kmin) {
    constexpr int SW = KeyTraits<KW>::SW;
    constexpr bool HASN = KeyTraits<KW>::HASN;
    score = 0; kmax = 0; kmin = 0;
    Query<SW> Q;
    query_setup<KW>(key, P, Q);
    if (!Q.ok) return;
    const int np = round_np(P);
    for (int p = 0; p < np; ++p) {
        if (!piece_needed(P, p, score)) break;                  // as the kernel: decided once per piece, before either strand
        for (int fw = 1; fw >= 0; --fw) {
            const Oriented<SW>& o = Q.q[fw];
            Strand<SW, HASN> S;
            strand_setup<SW, HASN>(o, fw, Q.l, Q.s, P, S);
            if (!S.ok) continue;
            int qs; uint64_t b0, b1;
            if (!piece_buckets_fwd<SW, HASN>(Q.q[1], Q.l, L.kb, fw, piece_cut(Q.span, p, np), piece_cut(Q.span, p + 1, np), qs, b0, b1)) continue;
            PieceAl A;
            piece_align<SW, HASN>(o, S, qs, fw, 0u, A);
            koff_range(L, b0, b1, A.lo, A.hi);
            const bool scan = A.hi - A.lo > 8;                  // as the kernel: long lists go through the pre-filter
            const PreAl R = pre_of(A);
            for (uint32_t c = A.lo; c < A.hi; ++c) {
                int sm, tm; uint32_t w;
                if (scan && !pre_pass(R, L.kpre[c], P.maxmm)) continue;
                int r = cand_check<SW>(L, A, L.kent[c], P, sm, tm, w);
                if (r == 2) r = window_ok<SW, HASN>(L, o, S.nsp, S.seedm, S.nN, S.nNseed, (int64_t)w, P, sm, tm) ? 1 : 0;
                if (r == 1) {
                    uint64_t x = hit_score(sm, tm, w, fw), y = hit_kmax(sm, w), z = hit_kmin(sm, w);
                    score = x > score ? x : score; kmax = y > kmax ? y : kmax; kmin = z > kmin ? z : kmin;
                }
            }
        }
    }
}
// One read in one round, as the kernels decide it: the probe path where the library and the read allow it, else
// the generic matcher.
template <int KW>
__host__ __device__ inline bool match_one(const uint64_t* key, const LibDev& L, const RoundDev& P, int32_t& ref, uint32_t& amb) {
    if (KW == 2 && lds_round(L, P)) {
        PosBest Bst;
        if (lds_one(key, L, P, Bst)) {
            if (!Bst.score) return false;
            ref = (int32_t)ref_of(L, (uint32_t)(Bst.score >> 1));
            amb = ref_of(L, (uint32_t)Bst.kmax) != ref_of(L, 0xFFFFFFFFu - (uint32_t)Bst.kmin);
            return true;
        }
    }
    if (probe_round(L, P)) {
        ProbeBest Bst;
        if (probe_one<KW>(key, L, P, Bst)) {
            if (!Bst.score) return false;
            ref = (int32_t)Bst.ref;
            amb = (uint32_t)Bst.rmax != 0xFFFFFFFFu - (uint32_t)Bst.rmin;
            return true;
        }
    }
    uint64_t score = 0, kmax = 0, kmin = 0;
    match_generic<KW>(key, L, P, score, kmax, kmin);
    if (!score) return false;
    ref = (int32_t)ref_of(L, (uint32_t)(score >> 1));
    amb = ref_of(L, (uint32_t)kmax) != ref_of(L, 0xFFFFFFFFu - (uint32_t)kmin);
    return true;
}

// ==========================================================================================
// Verbatim uniques: reads no key can hold -- longer than 124 nt, or holding characters other than A C G T N, which
// miRge.pl:306 keeps apart by their exact string (the collapse of these is a host-side string table, mirge_hip.hip).
// They are matched in the matcher's alphabet (a/c/g/t in either case = the base, every other character = N:
// hypothesis H4) with the same policy as every other read, serially, one thread per unique:
// candidates from the one-strand k-mer index by the pigeonhole pieces, each verified on the text word by word.
// Record of a unique in the pool: [ sq fwd (WS) | N mask fwd (WN) | sq reverse complement (WS) | N mask rc (WN) ],
// WS = ceil(len / 32) + 1, WN = ceil(len / 64) + 1 words (one word of zeros behind each part: every access below
// loads two neighbouring words).
MG_HD int verb_ws(int len) { return (len + 31) / 32 + 1; }
MG_HD int verb_wn(int len) { return (len + 63) / 64 + 1; }
MG_HD int verb_words(int len) { return 2 * (verb_ws(len) + verb_wn(len)); }
// host: the record of one read out of its characters
inline void verb_pack(const char* s, int len, uint64_t* rec) {
    const int WS = verb_ws(len), WN = verb_wn(len);
    for (int i = 0; i < verb_words(len); ++i) rec[i] = 0;
    uint64_t *sf = rec, *nf = rec + WS, *sr = rec + WS + WN, *nr = rec + 2 * WS + WN;
    for (int i = 0; i < len; ++i) {
        uint8_t ch = (uint8_t)s[i];
        if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);
        const int code = base_code(ch), j = len - 1 - i;
        if (code >= 0 && code < 4) { sf[i >> 5] |= (uint64_t)code << (2 * (i & 31)); sr[j >> 5] |= (uint64_t)(3 - code) << (2 * (j & 31)); }
        else { nf[i >> 6] |= 1ull << (i & 63); nr[j >> 6] |= 1ull << (j & 63); }
    }
}
MG_HD uint64_t lw_bits(const uint64_t* __restrict__ x, int bp) {
    const int wi = bp >> 6, bs = bp & 63;
    const uint64_t lo = x[wi], hi = x[wi + 1];
    return bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
}
struct LongRead { const uint64_t* sq[2]; const uint64_t* nm[2]; int ob[2]; int l, s, span; bool ok; };     // [1] forward, [0] reverse complement
MG_HD void long_setup(const uint64_t* __restrict__ rec, int len, const RoundDev& P, LongRead& R) {
    const int WS = verb_ws(len), WN = verb_wn(len);
    R.sq[1] = rec; R.nm[1] = rec + WS; R.sq[0] = rec + WS + WN; R.nm[0] = rec + 2 * WS + WN;
    R.ob[1] = P.trim5; R.ob[0] = P.trim3;                      // bowtie -5 / -3: the trimmed read starts trim3 nt into the reverse complement
    R.l = len - P.trim5 - P.trim3;
    R.ok = !(R.l <= 0 || R.l <= P.seedmm);
    R.s = P.seedlen < R.l ? P.seedlen : R.l;
    R.span = round_whole(P) ? R.l : R.s;
}
MG_HD bool long_window_ok(const LibDev& L, const LongRead& R, int fw, int64_t w, const RoundDev& P, int& sm_out, int& tm_out) {
    const int l = R.l;
    if (w < 0 || w + l > (int64_t)L.T) return false;
    const int sa = fw ? 0 : l - R.s, sb = fw ? R.s : l;        // the seed = the read's first s nt, wherever the strand puts them
    int tm = 0, sm = 0;
    for (int j = 0; 32 * j < l; ++j) {
        const uint64_t r = lw_bits(R.sq[fw], 2 * (R.ob[fw] + 32 * j));
        const uint64_t n = spread32((uint32_t)lw_bits(R.nm[fw], R.ob[fw] + 32 * j));
        const uint64_t x = text_bits64(L.text, w + 32 * j);
        uint64_t d = x ^ r;
        d = (((d | (d >> 1)) & M55) | n) & ones(2 * (l - 32 * j));             // a read N mismatches everything
        const uint64_t sd = (ones(2 * (sb - 32 * j)) ^ ones(2 * (sa - 32 * j))) & M55;
        tm += popc64(d); sm += popc64(d & sd);
        if (tm > P.maxmm) return false;
    }
    if (sm > P.seedmm) return false;
    for (int j = 0; 64 * j < l; ++j) if (bad_bits64(L.bad, w + 64 * j) & ones(l - 64 * j)) return false;      // reference N / boundary
    sm_out = sm; tm_out = tm;
    return true;
}
__host__ __device__ inline void match_long(const uint64_t* __restrict__ rec, int len, const LibDev& L, const RoundDev& P,
                                           uint64_t& score, uint64_t& kmax, uint64_t& kmin) {
    score = 0; kmax = 0; kmin = 0;
    LongRead R;
    long_setup(rec, len, P, R);
    if (!R.ok) return;
    const int np = round_np(P);
    for (int p = 0; p < np; ++p) {
        if (!piece_needed(P, p, score)) break;
        const int ps = piece_cut(R.span, p, np), pe = piece_cut(R.span, p + 1, np);
        for (int fw = 1; fw >= 0; --fw) {
            const int qs = fw ? ps : R.l - pe;
            const int kq = pe - ps < L.kb ? pe - ps : L.kb;
            if (kq <= 0) continue;
            if (lw_bits(R.nm[fw], R.ob[fw] + qs) & ones(kq)) continue;         // a piece holding a read N is not the exact one
            const uint64_t v = lw_bits(R.sq[fw], 2 * (R.ob[fw] + qs)) & ones(2 * kq);
            const uint32_t code = (uint32_t)(rev2(v) >> (64 - 2 * kq));
            const int sh = 2 * (L.kb - kq);
            uint32_t lo, hi;
            koff_range(L, (uint64_t)code << sh, ((uint64_t)code + 1) << sh, lo, hi);
            for (uint32_t c = lo; c < hi; ++c) {
                const uint32_t pos = L.kent[c].pos;
                if (pos < (uint32_t)qs) continue;
                const uint32_t w = pos - (uint32_t)qs;
                int sm, tm;
                if (!long_window_ok(L, R, fw, (int64_t)w, P, sm, tm)) continue;
                const uint64_t x = hit_score(sm, tm, w, fw), y = hit_kmax(sm, w), z = hit_kmin(sm, w);
                score = x > score ? x : score; kmax = y > kmax ? y : kmax; kmin = z > kmin ? z : kmin;
            }
        }
    }
}

// One cascade round over the unique table: eligibility (miRge.pl:382-393), match, mark
// (miRge.pl:445-446).  stats: [0] processed, [1] aligned (miRge.pl:409-413).
//
// Candidate lists differ by orders of magnitude between reads (a 4-nt seed piece of a 13-nt isomiR
// query against the miRNA library has ~280 candidates, a 14-nt piece ~1), so a thread-per-read loop
// runs at the pace of the unluckiest lane.  The block therefore works in three phases:
//   1. every thread prepares its read and, piece by piece, looks up the bucket range of both strands,
//      lays the piece out in context coordinates (PieceAl), checks short lists (<= thr entries)
//      itself and queues the long ones in LDS;
//   2. the waves of the block drain the queue, one entry at a time, 64 index entries per step, the
//      PieceAl being read back from LDS (same address in every lane: a broadcast);
//   3. every thread reads its result.
// A valid alignment is recorded by three LDS atomic max on its owner's keys (hit_score/kmax/kmin);
// valid alignments are rare next to candidates, so the atomics are off the critical path.  The kernel
// is instruction-issue bound (rocprof: SQ_ACTIVE_INST_ANY ~ 90 % of SIMD time), so the inner check
// is kept to one 16-byte load and ~20 VALU instructions.
// A survivor list as a launch reads / writes it.  Dense (K <= 1): p[0 .. n[0]).  Split (K > 1): K sub-lists, sub-list k =
// p[k * cap .. k * cap + n[k * LIST_NSTRIDE]) -- block b of the writing launch appends to sub-list b % K, so the K length
// words take the appends of a launch between them.  Why: every block of a round ends with ONE returning atomic on the
// list's length, and the device serves atomics on one address one after the other, ~12 ns each (they execute at the
// memory side, past the eight L2s): 43 K blocks of round 3 on one word are a 0.53 ms chain, whatever else the kernel does
// (profiles/r02/v4_phase_elimination.txt).  The reading launch maps block b to tile b / K of sub-list b % K, which
// costs it nothing; cap = (blocks of the writing launch that share a sub-list) x block size, so a sub-list cannot overrun.
constexpr int LIST_K = 64, LIST_NSTRIDE = 16;    // sub-lists of a split list; 64-bit words between their length words (128 B)
struct ListIn  { const uint32_t* p; const unsigned long long* n; uint32_t K, cap; };      // p == null: the whole table
struct ListOut { uint32_t* p; unsigned long long* n; uint32_t K, cap; };                  // p == null: no list is kept
constexpr int MB = 256;
#ifndef MG_QCAP
#define MG_QCAP 384
#endif
constexpr int QCAP = MG_QCAP;                        // LDS queue entries per block; a full queue falls back to the serial loop
constexpr int STAT_SLOTS = 256;                  // per round: [processed | aligned] x STAT_SLOTS words, 64 B apart

__device__ __forceinline__ void record_hit(unsigned long long* s_best, uint32_t owner, int sm, int tm, uint32_t w, int fw) {
    atomicMax(&s_best[owner], (unsigned long long)hit_score(sm, tm, w, fw));
    atomicMax(&s_best[MB + owner], (unsigned long long)hit_kmax(sm, w));
    atomicMax(&s_best[2 * MB + owner], (unsigned long long)hit_kmin(sm, w));
}
// rare path: the context did not cover the window; rebuild the owner's read from its key
template <int KW>
__device__ __forceinline__ void finish_on_text(const uint64_t* __restrict__ ukeys, uint32_t u, const LibDev& L, const RoundDev& P,
                                            int fw, uint32_t w, unsigned long long* s_best, uint32_t owner) {
    constexpr int SW = KeyTraits<KW>::SW;
    constexpr bool HASN = KeyTraits<KW>::HASN;
    uint64_t k[KW];
#pragma unroll
    for (int i = 0; i < KW; ++i) k[i] = ukeys[(uint64_t)u * KW + i];
    Query<SW> Q;
    query_setup<KW>(k, P, Q);
    Oriented<SW> o = fw ? Q.q[1] : Q.q[0];
    Strand<SW, HASN> S;
    strand_setup<SW, HASN>(o, fw, Q.l, Q.s, P, S);
    int sm, tm;
    if (window_ok<SW, HASN>(L, o, S.nsp, S.seedm, S.nN, S.nNseed, (int64_t)w, P, sm, tm)) record_hit(s_best, owner, sm, tm, w, fw);
}

#ifndef MG_MATCH_WAVES
#define MG_MATCH_WAVES 6
#endif
#ifndef MG_PF
#define MG_PF 2                                  // queue entries a wave has in flight in phase 2 (4 spills registers: slower)
#endif
template <int KW> struct MatchTile { static constexpr uint32_t N = KW == 2 ? 256u : 64u; };       // reads per block (match_block)
// one block's worth (MatchTile reads, block number bid of the launch's work) of a round
template <int KW, int RN>
__device__ __forceinline__ void match_block(const uint32_t bid, const uint32_t n, const uint64_t* __restrict__ ukeys, uint8_t* ann_round, int32_t* ann_ref,
              uint8_t* ann_amb, const LibDev& L, const RoundDev& P, int thr, unsigned long long* stats,
              const uint32_t* __restrict__ list, uint32_t* next_list, unsigned long long* next_n, const bool init) {
    // list (nullable) = the uniques no earlier round annotated, as left by the previous round's launch;
    // next_list (nullable) receives this round's survivors.  Dense lists keep the lanes busy in the
    // later rounds, where most of the table is already annotated.
    constexpr int SW = KeyTraits<KW>::SW;
    constexpr bool HASN = KeyTraits<KW>::HASN;
    __shared__ unsigned long long s_best[3 * MB];
    __shared__ PieceAl s_queue[QCAP];
    __shared__ uint32_t s_u[MB];
    __shared__ uint32_t s_qn, s_next, s_wcnt[MB / 64], s_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    // MT reads per block: the block's 256 threads for the 16-byte class; 64 for the 48-byte class -- its tables hold a few
    // ten thousand uniques (reads with N), which as 256-read tiles were ~150 blocks on a 256-CU chip, each a long serial
    // chain of phase 1 -> drain -> results (0.84 ms for 40 K uniques in the two-mismatch round); four times the blocks,
    // each draining a quarter of the lists with all of its waves
    constexpr uint32_t MT = MatchTile<KW>::N;
    const uint32_t i = bid * MT + tid;
    const uint32_t u = (tid < MT && i < n) ? (list ? list[i] : i) : EMPTY;
    s_u[tid] = u;
    bool active = false;
    uint64_t k[KW];
    // key and annotation state are fetched side by side (the key of an already annotated unique is a wasted load,
    // a load that waits for the state is a longer dependent chain for everyone else)
    uint8_t done_before = 0;
    if (u != EMPTY) {
        // the first launch that meets a unique (round 1's walk over the whole table; rounds 1 and 2 when each walks its
        // own length class, `init`) leaves the "not annotated" state behind instead of reading it: the host fills none
        // of the three arrays
        if (init) { ann_round[u] = 0; ann_ref[u] = -1; ann_amb[u] = 0; }
        else if (!list) done_before = ann_round[u];
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = ukeys[(uint64_t)u * KW + w];
    }
    const bool listed = u != EMPTY && done_before == 0;
    if (listed) {
        int len = key_len<KW>(k);
        active = !(P.len_lt && !(len < P.len_lt)) && !(P.len_gt && !(len > P.len_gt));
    }
    if (!__syncthreads_or(active) && !next_list) return;
    s_best[tid] = 0; s_best[MB + tid] = 0; s_best[2 * MB + tid] = 0;
    if (tid == 0) { s_qn = 0; s_next = 0; }
    __syncthreads();

#ifdef MG_PHASE_PROF
    const long long t0_ = clock64();
#endif
    // ---- phase 1
    Query<SW> Q;
    if (!active) {
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = 0;                 // an empty read: Q.ok comes out false
    }
    query_setup_fwd<KW>(k, P, Q);
    Q.ok = Q.ok && active;
    const int np = round_np(P);
    // The kernel is VALU-issue bound (~1500 instructions per read in this phase before this ordering), and most
    // lookups end in an empty bucket: so the bucket ranges of both strands are fetched first, and the strand
    // constants + context layout (the expensive part) are computed only for a piece that has candidates.
    bool sok[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        sok[f] = Q.ok;       // (the N counts of a read and of its seed are the same on both strands)
        if (HASN) { Strand<SW, HASN> T; strand_setup<SW, HASN>(Q.q[1], 1, Q.l, Q.s, P, T); sok[f] = sok[f] && T.ok; }
    }
    for (int p = 0; p < np; ++p) {
        const int ps = piece_cut(Q.span, p, np), pe = piece_cut(Q.span, p + 1, np);
        uint32_t lo[2], hi[2]; int qs[2];
        // what this thread checked itself is in s_best[tid] by now (queued lists are not: they can only add hits)
        const bool need = p == 0 || piece_needed(P, p, s_best[tid]);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            uint64_t b0 = 0, b1 = 0;
            lo[f] = 0; hi[f] = 0; qs[f] = 0;
            if (need && sok[f] && piece_buckets_fwd<SW, HASN>(Q.q[1], Q.l, L.kb, f, ps, pe, qs[f], b0, b1)) koff_range(L, b0, b1, lo[f], hi[f]);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const uint32_t cnt = hi[f] - lo[f];
            if (cnt == 0) continue;
            Oriented<SW> o = Q.q[1];
            if (!f) revcomp<SW>(Q.q[1], o);
            Strand<SW, HASN> S;
            strand_setup<SW, HASN>(o, f, Q.l, Q.s, P, S);
            PieceAl A;
            piece_align<SW, HASN>(o, S, qs[f], f, tid, A);
            A.lo = lo[f]; A.hi = hi[f];
            uint32_t e = QCAP;
            if (cnt > (uint32_t)thr) e = atomicAdd(&s_qn, 1u);
            if (e < (uint32_t)QCAP) { s_queue[e] = A; continue; }
            for (uint32_t c = A.lo; c < A.hi; ++c) {
                int sm, tm; uint32_t w;
                int r = cand_check<SW>(L, A, L.kent[c], P, sm, tm, w);
                if (r == 1) record_hit(s_best, tid, sm, tm, w, f);
                else if (r == 2) finish_on_text<KW>(ukeys, u, L, P, f, w, s_best, tid);
            }
        }
    }
    __syncthreads();
#ifdef MG_PHASE_PROF
    const long long t1_ = clock64();
#endif

    // ---- phase 2: drain the queue.  A wave takes MG_PF entries at a time and issues the first 64 pre-filter loads
    // of each before it looks at any of them: with one entry (one load) in flight per wave the drain ran at the
    // pace of memory latency (~3 G entries/s chip-wide), not of bandwidth or issue.
    const uint32_t qn = s_qn < (uint32_t)QCAP ? s_qn : (uint32_t)QCAP;
    for (;;) {
        uint32_t e0 = 0;
        if (lane == 0) e0 = atomicAdd(&s_next, (uint32_t)MG_PF);
        e0 = __shfl(e0, 0);
        if (e0 >= qn) break;
        // while the loads are in flight a list costs four registers (bounds + the loaded word); the prepared piece is
        // read back from LDS (a broadcast) only when its words have arrived: no spills at 80 VGPRs
        uint64_t pv[MG_PF]; uint32_t lo[MG_PF], hi[MG_PF];
#pragma unroll
        for (int j = 0; j < MG_PF; ++j) {
            const uint32_t e = e0 + j < qn ? e0 + j : qn - 1;       // clamp: a valid entry, ignored below
            lo[j] = s_queue[e].lo; hi[j] = e0 + j < qn ? s_queue[e].hi : lo[j];
            const uint32_t c = lo[j] + lane;
            pv[j] = c < hi[j] ? L.kpre[c] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < MG_PF; ++j) {
            if (e0 + j >= qn) break;
            uint32_t c = lo[j] + lane;
            const PreAl Rj = pre_of(s_queue[e0 + j]);
            bool pass = c < hi[j] && pre_pass(Rj, pv[j], P.maxmm);
            for (;;) {
                if (pass) {
                    const PieceAl A = s_queue[e0 + j];
                    const uint32_t owner = A.meta & 255; const int f = (int)((A.meta >> 8) & 1);
                    int sm, tm; uint32_t w;
                    int r = cand_check<SW>(L, A, L.kent[c], P, sm, tm, w);
                    if (r == 1) record_hit(s_best, owner, sm, tm, w, f);
                    else if (r == 2) finish_on_text<KW>(ukeys, s_u[owner], L, P, f, w, s_best, owner);
                }
                c += 64;
                if (!__any(c < hi[j])) break;                   // lists longer than 64: the remaining strides, one load each
                pass = c < hi[j] && pre_pass(Rj, L.kpre[c < hi[j] ? c : lo[j]], P.maxmm);
            }
        }
    }
    __syncthreads();

#ifdef MG_PHASE_PROF
    if (tid == 0) {
        const long long t2_ = clock64();
        uint32_t slot = bid & (STAT_SLOTS - 1);
        atomicAdd(&stats[(size_t)slot * 8 + 1], (unsigned long long)(t1_ - t0_));
        atomicAdd(&stats[(size_t)slot * 8 + 2], (unsigned long long)(t2_ - t1_));
        atomicAdd(&stats[(size_t)slot * 8 + 3], (unsigned long long)qn);
        atomicAdd(&stats[(size_t)slot * 8 + 4], 1ull);
    }
#endif
    // ---- phase 3
    bool hit = false;
    if (Q.ok) {
        unsigned long long score = s_best[tid];
        if (score) {
            hit = true;
            ann_round[u] = (uint8_t)(RN + 1);
            const uint32_t r = ref_of(L, (uint32_t)(score >> 1));
            ann_ref[u] = (int32_t)r;
            // ambiguous = the best stratum spans more than one reference; its extremes usually lie in r itself
            const uint32_t pmax = (uint32_t)s_best[MB + tid], pmin = 0xFFFFFFFFu - (uint32_t)s_best[2 * MB + tid];
            const uint32_t r0 = L.ref_start[r], r1 = L.ref_start[r + 1];
            bool amb = false;
            if (pmin < r0 || pmax >= r1) amb = ref_of(L, pmax) != ref_of(L, pmin);
            ann_amb[u] = (uint8_t)amb;
        }
    }
    // counters are spread over STAT_SLOTS cache lines: one hot word would cost ~12 ns per wave
    unsigned long long am = __ballot(active), hm = __ballot(hit);
    if (lane == 0) {
        uint32_t slot = (bid * 4 + (tid >> 6)) & (STAT_SLOTS - 1);
        if (am) atomicAdd(&stats[(size_t)slot * 8], (unsigned long long)__popcll(am));
        if (hm) atomicAdd(&stats[(size_t)(STAT_SLOTS + slot) * 8], (unsigned long long)__popcll(hm));
    }
    // survivors -> next round's list (one atomic per block)
    if (next_list) {
        const bool surv = listed && !hit;
        const unsigned long long sm_ = __ballot(surv);
        if (lane == 0) s_wcnt[tid >> 6] = (uint32_t)__popcll(sm_);
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < MB / 64; ++w) { uint32_t c = s_wcnt[w]; s_wcnt[w] = tot; tot += c; }
            s_base = tot ? (uint32_t)atomicAdd(next_n, (unsigned long long)tot) : 0u;
        }
        __syncthreads();
        if (surv) next_list[s_base + s_wcnt[tid >> 6] + (uint32_t)__popcll(sm_ & ((1ull << lane) - 1))] = u;
    }
}

// The kernel: blocks walk the launch's work with a grid stride, so a launch over a list whose length only the device
// knows (what the probe kernel hands on: usually a handful of reads) can be a small grid instead of one block per
// MB uniques of the table that leave at once.
template <int KW, int RN>
__global__ void __launch_bounds__(MB, MG_MATCH_WAVES)
k_match_round(const uint64_t* __restrict__ ukeys, uint32_t U, uint8_t* ann_round, int32_t* ann_ref,
              uint8_t* ann_amb, LibDev L, RoundDev P, int thr, unsigned long long* stats,
              ListIn in, const uint32_t* __restrict__ list2, const unsigned long long* __restrict__ list2_n,
              uint32_t* next_list, unsigned long long* next_n, bool init,
              const unsigned long long* __restrict__ U_dev, uint32_t U0) {
    // U_dev (nullable): the table's size is still on the device -- U0 uniques + the new ones of a collapse whose packed
    // rank counter (new uniques in its high half) the host has not read yet; U is then an upper bound (grid sizing)
    // in: the launch's work (dense, split, or the whole table); list2 (nullable): a second, dense list behind it (what
    // earlier launches of this kernel left of the reads the probe kernel had handed them).  Survivors go to ONE dense list.
    if (U_dev) { const uint32_t ud = U0 + (uint32_t)(*U_dev >> 32); U = ud < U ? ud : U; }      // never beyond the bound the launch and its buffers were sized for
    if (in.K > 1) {
        constexpr uint32_t MT = MatchTile<KW>::N;
        const uint32_t per = in.cap / MT;                  // tiles a sub-list can hold (cap is a multiple of the block size)
        for (uint32_t t = blockIdx.x; t < in.K * per; t += gridDim.x) {
            const uint32_t k = t % in.K, j = t / in.K;
            const uint32_t nk = (uint32_t)in.n[(size_t)k * LIST_NSTRIDE];
            if ((uint64_t)j * MT >= nk) continue;
            match_block<KW, RN>(j, nk, ukeys, ann_round, ann_ref, ann_amb, L, P, thr, stats, in.p + (size_t)k * in.cap, next_list, next_n, init);
            __syncthreads();                               // the block's LDS state is reused by its next piece of work
        }
    } else {
        const uint32_t n = (in.p && in.n) ? (uint32_t)*in.n : U;          // (a list without a length word is as long as the table: EMPTY = a hole)
        for (uint32_t bid = blockIdx.x; (uint64_t)bid * MatchTile<KW>::N < n; bid += gridDim.x) {
            match_block<KW, RN>(bid, n, ukeys, ann_round, ann_ref, ann_amb, L, P, thr, stats, in.p, next_list, next_n, init);
            __syncthreads();
        }
    }
    if (list2) {
        const uint32_t n2 = (uint32_t)*list2_n;
        for (uint32_t bid = blockIdx.x; (uint64_t)bid * MatchTile<KW>::N < n2; bid += gridDim.x) {
            match_block<KW, RN>(bid, n2, ukeys, ann_round, ann_ref, ann_amb, L, P, thr, stats, list2, next_list, next_n, false);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// The probe path as a kernel (see probe_one for the serial form).  One thread per unique:
//   1. key, eligibility, the read's two orientations, its lookups (pieces or a block scheme's shapes);
//   2. two lookups at a time: both occupancy words, then both slots -- two dependent steps, every lane with
//      several loads in flight;
//   3. a slot that holds its bucket's only entry is judged on the spot (own results stay in registers); a bucket
//      list is queued in LDS: up to 16 entries in chunks of four, longer ones in chunks of 64;
//   4. reads with more than two lookups (block schemes, three pieces) repeat 2-3 for the next pair, unless what they
//      have found themselves already rules those lookups out;
//   5. the block drains the queues -- sixteen 4-entry chunks, or one 64-entry chunk, per wave step (lane = chunk x
//      entry), the owner's read coming from LDS; results meet the owners' through LDS atomic max;
//   6. results, survivors -> next round's list, unserved reads -> the fallback list of the generic kernel.
constexpr int PB = 256;                          // threads per block
// queued chunks per block: rounds of two-mismatch schemes (NS = 3: 6 lookups of 8- or 6-nt specificity) fill them,
// the others hardly use them
// (one size for every instantiation: with 320 / 32 the rounds without two-mismatch schemes overflowed on libraries with
// repeated k-mers -- tRNA-like families put 10-50 entries behind most lookups of their reads, > 16 such reads per block
// filled the 32 long chunks, and 2.7 % of the reads went to the generic kernel, which then took 3 ms: bench.py `repeats`)
#ifndef MG_PQ4
#define MG_PQ4 768
#endif
#ifndef MG_PQ64
#define MG_PQ64 256
#endif
template <int NS> struct ProbeQ { static constexpr int Q4 = MG_PQ4, Q64 = MG_PQ64; };
#ifndef MG_PROBE_WAVES
#define MG_PROBE_WAVES 7                        // the scheme-carrying instantiations: 72 registers and 8-12 B of scratch (6 waves: none; 8: round 3 0.99 ms)
#endif
#ifndef MG_PROBE_WAVES3
#define MG_PROBE_WAVES3 6                       // several schemes, of unequal blocks (the two-mismatch round): 56 B of scratch at 7 waves, none at 6 -- round 5 0.47 -> 0.40 ms
#endif
#ifndef MG_PROBE_WAVES0
#define MG_PROBE_WAVES0 7                       // the scheme-free instantiation (rounds 1 and 4 of the reference cascade): 72 registers, no scratch (8 waves: 64 registers + 20 B of scratch since the long-list fold, rounds 1 / 4 0.127 / 0.312 ms against 0.121 / 0.290; 6 waves: 0.14 / 0.30)
#endif
// chunk: lo = first entry; meta = owner (0-7) | entries - 1 (8-13) | ps (14-19) | K (20-24) | written (31)
MG_HD uint32_t chunk_meta(uint32_t owner, uint32_t cnt, int ps, int K) {
    return owner | ((cnt - 1) << 8) | ((uint32_t)ps << 14) | ((uint32_t)K << 20) | 0x80000000u;
}
// KW = 6: the 48-byte class (reads with N, or longer than 56 nt: those go on the fb list) through the same tables
template <int RN, int NS, int KW = 2>
__global__ void __launch_bounds__(PB, NS == 0 ? MG_PROBE_WAVES0 : (NS >= 3 ? MG_PROBE_WAVES3 : MG_PROBE_WAVES))
k_probe_round(const uint64_t* __restrict__ ukeys, uint32_t U, uint8_t* ann_round, int32_t* ann_ref, uint8_t* ann_amb,
              LibDev L, RoundDev P, RoundSch RS, unsigned long long* stats,
              ListIn in, ListOut out, uint32_t* fb_list, unsigned long long* fb_n, bool init,
              const unsigned long long* __restrict__ U_dev, uint32_t U0) {
    constexpr int PQ4 = ProbeQ<NS>::Q4, PQ64 = ProbeQ<NS>::Q64;
    if (U_dev) { const uint32_t ud = U0 + (uint32_t)(*U_dev >> 32); U = ud < U ? ud : U; }      // as in k_match_round
    __shared__ unsigned long long s_best[3 * PB];
    __shared__ uint64_t s_r0[PB], s_r1[PB];
    __shared__ uint32_t s_ls[PB];                // l | s << 8
    __shared__ uint64_t s_nm[KW == 6 ? PB : 1];  // 48-byte class: the reads' N masks
    __shared__ uint32_t s_q4lo[PQ4], s_q4meta[PQ4], s_q64lo[PQ64], s_q64meta[PQ64];
    __shared__ uint32_t s_qn4, s_qn64, s_wcnt[2][PB / 64], s_base[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    // this block's tile of the work: tile blockIdx.x of a dense list (or of the table), tile b / K of sub-list b % K of a split one
    const uint32_t* __restrict__ list = in.p;
    uint32_t n, i0;
    if (in.K > 1) {
        const uint32_t sk = blockIdx.x % in.K;
        n = (uint32_t)in.n[(size_t)sk * LIST_NSTRIDE]; i0 = (blockIdx.x / in.K) * PB; list += (size_t)sk * in.cap;
    } else {
        n = (list && in.n) ? (uint32_t)*in.n : U; i0 = blockIdx.x * PB;
    }
    if (i0 >= n) return;
    uint32_t* next_list = out.p;
    unsigned long long* next_n = out.n;
    uint32_t next_cap = 0xFFFFFFFFu;
    if (out.K > 1) {
        const uint32_t ok = blockIdx.x % out.K;
        next_list += (size_t)ok * out.cap; next_n += (size_t)ok * LIST_NSTRIDE; next_cap = out.cap;
    }
    const uint32_t i = i0 + tid;
    const uint32_t u = i < n ? (list ? list[i] : i) : EMPTY;
    uint64_t k[KW];
#pragma unroll
    for (int w = 0; w < KW; ++w) k[w] = 0;
    uint8_t done_before = 0;
    if (u != EMPTY) {
        if (init) { ann_round[u] = 0; ann_ref[u] = -1; ann_amb[u] = 0; }       // as in k_match_round
        else if (!list) done_before = ann_round[u];
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = ukeys[(uint64_t)u * KW + w];
    }
    const bool listed = u != EMPTY && done_before == 0;
    bool active = false;
    if (listed) {
        const int len = key_len<KW>(k);
        active = !(P.len_lt && !(len < P.len_lt)) && !(P.len_gt && !(len > P.len_gt));
    }
    if (!__syncthreads_or(active) && !next_list) return;
    s_best[tid] = 0; s_best[PB + tid] = 0; s_best[2 * PB + tid] = 0;
    for (uint32_t j = tid; j < (uint32_t)PQ4; j += PB) s_q4meta[j] = 0;       // a chunk that is never written (queue overflow) stays empty
    for (uint32_t j = tid; j < (uint32_t)PQ64; j += PB) s_q64meta[j] = 0;
    if (tid == 0) { s_qn4 = 0; s_qn64 = 0; }
    if (!active) {
#pragma unroll
        for (int w = 0; w < KW; ++w) k[w] = 0;                  // an empty read: ok comes out false
    }
    ProbeRead R;
    probe_setup<NS, KW>(k, L, P, RS, R);
    R.ok = R.ok && active;
    const bool run = R.ok && R.served;
    bool fb = R.ok && !R.served;                                // -> the generic kernel
    s_r0[tid] = R.r[0]; s_r1[tid] = R.r[1]; s_ls[tid] = (uint32_t)(R.l & 255) | ((uint32_t)(R.s & 255) << 8);
    if (KW == 6) s_nm[tid] = R.nraw;
    __syncthreads();

    ProbeBest Bst; Bst.score = 0; Bst.rmax = 0; Bst.rmin = 0; Bst.ref = 0;      // what this thread finds itself
    const int K = tab_K(R.tab, L);
    // a slot of one lookup: judge the single entry / queue the list
    auto take = [&](const PEnt& e, int ps) {
        uint32_t lo = 0, hi = 0;
        const int kind = slot_kind(e, lo, hi);
        if (kind == 2) {
            int f, sm, tm; uint32_t w, ref;
            if (probe_verify<KW == 6>(P, R, K, ps, e, f, sm, tm, w, ref)) probe_record(Bst, sm, tm, w, f, ref);
        } else if (kind == 1) {
            const uint32_t len = hi - lo;
            if (len > (uint32_t)PROBE_MAX_LIST) { fb = true; return; }
            if (len <= 16) {
                const uint32_t nch = (len + 3) >> 2;
                const uint32_t at = atomicAdd(&s_qn4, nch);
                if (at + nch > (uint32_t)PQ4) { fb = true; return; }
                for (uint32_t c = 0; c < nch; ++c) {
                    s_q4lo[at + c] = lo + 4 * c;
                    s_q4meta[at + c] = chunk_meta(tid, len - 4 * c < 4 ? len - 4 * c : 4, ps, K);
                }
            } else {
                const uint32_t nch = (len + 63) >> 6;
                const uint32_t at = atomicAdd(&s_qn64, nch);
                if (at + nch > (uint32_t)PQ64) { fb = true; return; }
                for (uint32_t c = 0; c < nch; ++c) {
                    s_q64lo[at + c] = lo + 64 * c;
                    s_q64meta[at + c] = chunk_meta(tid, len - 64 * c < 64 ? len - 64 * c : 64, ps, K);
                }
            }
        }
    };
    // lookups two at a time: both occupancy words, then both slots, then the candidates.  The first pair is
    // unconditional; later ones are skipped where what this thread has found itself already rules them out.
    const bool ubits = tab_bits(R.tab);
#ifndef MG_PROBE_STEP
#define MG_PROBE_STEP 2
#endif
    constexpr int PW = MG_PROBE_STEP;                          // lookups in flight per thread
    for (int i0 = 0; i0 < PROBE_MAX_LOOK; i0 += PW) {
        uint32_t slot[PW], b[PW]; int ps[PW]; bool have[PW];
        bool any = false;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            have[j] = run && !fb && i0 + j < R.nlook && probe_needed(R, P, i0 + j, Bst.score);
            if (KW == 6) have[j] = have[j] && !probe_lookup_n(R, L, i0 + j);
            slot[j] = 0; b[j] = 0; ps[j] = 0;
            if (have[j]) probe_lookup<NS == 1>(R, L, i0 + j, slot[j], b[j], ps[j]);
            any = any || have[j];
        }
        if (i0 && !__any(any)) break;
        uint32_t wd[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) wd[j] = (have[j] && ubits) ? L.sbits[b[j] >> 5] : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PW; ++j) have[j] = have[j] && ((wd[j] >> (b[j] & 31)) & 1);
        PEnt e[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) { e[j].pos = SLOT_EMPTY; if (have[j]) e[j] = load_pent(L.slots, slot[j]); }
#pragma unroll
        for (int j = 0; j < PW; ++j) if (have[j]) take(e[j], ps[j]);
    }
    __syncthreads();

    // ---- drain: lane = (chunk, entry); every chunk's owner is told through LDS
    // one list entry against its owner's read: the three order keys of the alignment, 0 = none.  The 64 lanes of a step
    // of the long queue share ONE owner: they fold their keys by shuffles first and one lane tells the owner -- on
    // libraries with repeated sequences (families of near-identical members) most entries of a list verify, and 64 LDS
    // atomics on one word are served one after the other (bench.py `repeats`: round 3 1.81 -> 0.96 ms).
    auto judge = [&](bool go, uint32_t meta, uint32_t at, unsigned long long& x, unsigned long long& y, unsigned long long& z) {
        x = 0; y = 0; z = 0;
        if (!go) return;
        const uint32_t owner = meta & 255;
        ProbeRead O;
        O.r[0] = s_r0[owner]; O.r[1] = s_r1[owner];
        O.l = (int)(s_ls[owner] & 255); O.s = (int)((s_ls[owner] >> 8) & 255);
        O.nraw = KW == 6 ? s_nm[owner] : 0ull;
        const PEnt e = load_pent(L.pent, at);
        probe_masks(O);
        int f, sm, tm; uint32_t w, ref;
        if (probe_verify<KW == 6>(P, O, (int)((meta >> 20) & 31), (int)((meta >> 14) & 63), e, f, sm, tm, w, ref)) {
            x = hit_score(sm, tm, w, f); y = hit_kmax(sm, ref); z = hit_kmin(sm, ref);
        }
    };
    auto fold_max = [&](unsigned long long& v, int width) {       // max over aligned groups of `width` lanes
        for (int o = 1; o < width; o <<= 1) { const unsigned long long t = __shfl_xor(v, o); v = t > v ? t : v; }
    };
    auto tell = [&](uint32_t owner, unsigned long long x, unsigned long long y, unsigned long long z) {
        if (!x) return;
        atomicMax(&s_best[owner], x); atomicMax(&s_best[PB + owner], y); atomicMax(&s_best[2 * PB + owner], z);
    };
    {
        const uint32_t qn = s_qn4 < (uint32_t)PQ4 ? s_qn4 : (uint32_t)PQ4;
        for (uint32_t base = (tid >> 6) * 16; base < qn; base += (PB / 64) * 16) {
            const uint32_t j = base + (lane >> 2), ei = lane & 3;
            const uint32_t meta = j < qn ? s_q4meta[j] : 0u;
            if ((meta >> 31) && ei <= ((meta >> 8) & 63)) {       // (short lists: at most 4 lanes of a step share an owner -- told directly)
                const uint32_t owner = meta & 255;
                ProbeRead O;
                O.r[0] = s_r0[owner]; O.r[1] = s_r1[owner];
                O.l = (int)(s_ls[owner] & 255); O.s = (int)((s_ls[owner] >> 8) & 255);
                O.nraw = KW == 6 ? s_nm[owner] : 0ull;
                const PEnt e = load_pent(L.pent, s_q4lo[j] + ei);
                probe_masks(O);
                int f, sm, tm; uint32_t w, ref;
                if (probe_verify<KW == 6>(P, O, (int)((meta >> 20) & 31), (int)((meta >> 14) & 63), e, f, sm, tm, w, ref)) {
                    atomicMax(&s_best[owner], (unsigned long long)hit_score(sm, tm, w, f));
                    atomicMax(&s_best[PB + owner], (unsigned long long)hit_kmax(sm, ref));
                    atomicMax(&s_best[2 * PB + owner], (unsigned long long)hit_kmin(sm, ref));
                }
            }
        }
        // a 64-entry chunk has ONE owner: the wave takes the owner's read into scalar registers (readfirstlane) and
        // walks the list through its 8-byte pre-filter words first -- a lane's share is one 8-byte load, XOR and a
        // popcount; the 32-byte entry is fetched only for a candidate the word does not rule out (the drain of the
        // two-mismatch round was bound by the lines its ~35-entry lists span, not by arithmetic or latency: scalar
        // owner data alone, and several chunks per step, both measured flat)
        const uint32_t qm0 = s_qn64 < (uint32_t)PQ64 ? s_qn64 : (uint32_t)PQ64;
        const uint32_t qm = (uint32_t)__builtin_amdgcn_readfirstlane((int)qm0);
        for (uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); j < qm; j += PB / 64) {
            const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q64meta[j]);
            if (!(meta >> 31)) continue;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_q64lo[j]);
            const uint32_t owner = meta & 255;
            const uint64_t r0 = s_r0[owner], r1 = s_r1[owner];
            const uint64_t u0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r0) |
                                ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r0 >> 32)) << 32);
            const uint64_t u1 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r1) |
                                ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(r1 >> 32)) << 32);
            const uint32_t ls = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ls[owner]);
            uint64_t rr, pm;
            pre_query(u0, u1, (int)(ls & 255), (int)((meta >> 14) & 63), rr, pm);
            unsigned long long x, y, z;
            judge(lane <= ((meta >> 8) & 63) && pre_mism(rr, pm, L.ppre[lo + lane]) <= P.maxmm, meta, lo + lane, x, y, z);
            if (__any(x != 0)) {
                // the wave's best alignment, then the largest and the smallest reference among the lanes of ITS stratum
                // (what the maxima of hit_kmax / hit_kmin over all lanes come to): one 64-bit and two 32-bit folds
                unsigned long long bx = x;
                fold_max(bx, 64);
                const bool in = x != 0 && (x >> 36) == (bx >> 36);
                uint32_t rmx = in ? (uint32_t)y : 0u, rmn = in ? (uint32_t)z : 0u;       // (z holds 0xFFFFFFFF - ref)
                for (int o = 1; o < 64; o <<= 1) { const uint32_t a = __shfl_xor(rmx, o), b = __shfl_xor(rmn, o); rmx = a > rmx ? a : rmx; rmn = b > rmn ? b : rmn; }
                if (lane == 0) tell(owner, bx, ((bx >> 36) << 32) | rmx, ((bx >> 36) << 32) | rmn);
            }
        }
    }
    __syncthreads();

    // ---- results
    bool hit = false;
    if (run && !fb) {
        const unsigned long long a = s_best[tid], b2 = s_best[PB + tid], c2 = s_best[2 * PB + tid];
        uint32_t ref = Bst.ref;
        if (a > Bst.score) { Bst.score = a; ref = ref_of(L, (uint32_t)(a >> 1)); }      // the representative came out of a list: its reference from its position
        Bst.rmax = b2 > Bst.rmax ? b2 : Bst.rmax; Bst.rmin = c2 > Bst.rmin ? c2 : Bst.rmin;
        if (Bst.score) {
            hit = true;
            ann_round[u] = (uint8_t)(RN + 1);
            ann_ref[u] = (int32_t)ref;
            ann_amb[u] = (uint8_t)((uint32_t)Bst.rmax != 0xFFFFFFFFu - (uint32_t)Bst.rmin);
        }
    }
    // unserved reads are counted (and listed as survivors, if they stay unaligned) by the generic kernel that takes them
    const unsigned long long am = __ballot(active && !fb), hm = __ballot(hit);
    if (lane == 0) {
        const uint32_t slot = (blockIdx.x * 4 + (tid >> 6)) & (STAT_SLOTS - 1);
        if (am) atomicAdd(&stats[(size_t)slot * 8], (unsigned long long)__popcll(am));
        if (hm) atomicAdd(&stats[(size_t)(STAT_SLOTS + slot) * 8], (unsigned long long)__popcll(hm));
    }
    const bool surv = next_list != nullptr && listed && !hit && !fb;
    const unsigned long long sv = __ballot(surv), fv = __ballot(fb);
    if (lane == 0) { s_wcnt[0][tid >> 6] = (uint32_t)__popcll(sv); s_wcnt[1][tid >> 6] = (uint32_t)__popcll(fv); }
    __syncthreads();
    if (tid < 2) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < PB / 64; ++w) { const uint32_t c = s_wcnt[tid][w]; s_wcnt[tid][w] = tot; tot += c; }
        s_base[tid] = tot ? (uint32_t)atomicAdd(tid ? fb_n : next_n, (unsigned long long)tot) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1;
    if (surv) {
        const uint32_t at = s_base[0] + s_wcnt[0][tid >> 6] + (uint32_t)__popcll(sv & below);
        if (at < next_cap) next_list[at] = u;               // (cannot fail: cap = the blocks that share the sub-list x PB)
    }
    if (fb) fb_list[s_base[1] + s_wcnt[1][tid >> 6] + (uint32_t)__popcll(fv & below)] = u;
}

// ------------------------------------------------------------------------------------------
// The LDS path as a kernel (see lds_one for the serial form): a round against a library whose 2-bit text and N bitmap
// fit the LDS (BASELINE.json's north_star: "library resident in LDS/L2" -- the miRNA library is 18 KB + 9 KB).
//   * Persistent blocks: each stages text + bitmap into LDS ONCE, then its waves stride over 64-read tiles of the
//     launch's work.  After the staging there is no block barrier: a wave is a worker of its own, with its own queues
//     and result words in LDS, so waves never wait for each other's memory round trips.
//   * Per tile: key -> eligibility -> the read's lookups (probe_setup: pieces, or a block scheme's shapes) -> every
//     occupancy word, then every 8-byte slot of the read at once (all in flight) -> a bucket's only entry is verified
//     on the spot; bucket lists are queued: up to LDS_SHORT entries in the short queue (drained 64 / LDS_SHORT lists per
//     wave step), longer ones in the long queue (64 entries per step, ONE owner, whose read sits in scalar registers).
//     A list entry is 4 bytes (text position | strand), read coalesced; verification = 5 LDS reads + XOR / popcount.
//   * Results meet the owner through LDS atomic max on the three order keys of the generic matcher (positions).
// What the path cannot serve (no fitting scheme, a bucket of more than LDS_MAX_LIST entries) goes on the `fb` list for
// the generic kernel, as on the probe path.  Lists in / out as k_probe_round (a tile here is 64 reads).

constexpr int LDS_SHORT = 16;                    // entries of a "short" list
constexpr int LDS_SQ = 128, LDS_LQ = 64;         // queue capacities per wave (each holds at least one lookup step's lists: 64)
#ifndef MG_LDS_LB
#define MG_LDS_LB 8
#endif
#ifndef MG_LDS_SB
#define MG_LDS_SB 4
#endif
constexpr int LDS_LB = MG_LDS_LB, LDS_SB = MG_LDS_SB;      // long lists / short-queue steps whose entries a drain has in flight at once
// The three order keys of an alignment in 32 bits each (text positions of such a library fit 24 bits): same order as
// hit_score / hit_kmax / hit_kmin.
MG_HD uint32_t sc32(int sm, int tm, uint32_t w, int f) { return ((uint32_t)(4 - sm) << 28) | ((uint32_t)tm << 26) | (w << 1) | (uint32_t)f; }
MG_HD uint32_t kx32(int sm, uint32_t w) { return ((uint32_t)(4 - sm) << 28) | w; }
MG_HD uint32_t kn32(int sm, uint32_t w) { return ((uint32_t)(4 - sm) << 28) | (0xFFFFFFu - w); }
MG_HD uint64_t sc32_as_score(uint32_t x) { return x ? ((uint64_t)(x >> 28) << 36) | 1ull : 0ull; }      // what probe_needed looks at: the stratum
struct LdsWave {
    uint32_t best[3][64];                        // score / kmax / kmin of what the drains found, per owner lane
    uint64_t r0[64], r1[64];                     // the lanes' reads
    uint32_t ls[64];                             // l | s << 8
    uint32_t cl[64];                             // first 16 nt of their reverse complements (the drains' pre-filter)
    uint64_t sq[LDS_SQ], lq[LDS_LQ];             // list descriptors: lo (0-23) | n (24-39) | owner (40-45) | ps (46-51) | K (52-57)
};
MG_HD uint64_t lds_desc(uint32_t lo, uint32_t n, uint32_t owner, int ps, int K) {
    return (uint64_t)lo | ((uint64_t)n << 24) | ((uint64_t)owner << 40) | ((uint64_t)ps << 46) | ((uint64_t)K << 52);
}
constexpr uint32_t LDS_MAX_ENTRIES = 1u << 24;   // entries of all compact tables of a library (descriptor: 24 bits)
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in program order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
// The drains' pre-filter: the first min(l, 16) nt of the alignment an entry stands for -- one 8-byte LDS read, a funnel
// shift, XOR and a popcount, ~15 vector instructions.  Nearly every entry of a short-key bucket fails here (a 6-nt bucket
// of the miRNA library holds ~36 entries, of which ~0.04 % pass); the full check (pos_verify, ~10x the instructions) runs
// for the others only.  off_f / off_r = where the looked-up window starts in the read / in its reverse complement;
// q_f / q_r = the first 16 nt of either; m32 = mask of min(l, 16) nt.
__device__ __forceinline__ bool lds_prefilter(const uint32_t* text32, uint32_t T, uint32_t e, int l, int off_f, int off_r,
                                              uint32_t q_f, uint32_t q_r, uint32_t m32, int maxmm) {
    const bool f = !(e >> 31);
    const int w = (int)(e & POS_MASK) - (f ? off_f : off_r);
    const bool ok = w >= 0 && (uint32_t)(w + l) <= T;
    const uint32_t wc = ok ? (uint32_t)w : 0u;
    const uint32_t lo = text32[wc >> 4], hi = text32[(wc >> 4) + 1];
    const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, (wc & 15) * 2);
    uint32_t d = x ^ (f ? q_f : q_r);
    d = (d | (d >> 1)) & m32;
    return ok && __popc(d) <= maxmm;
}
MG_HD uint32_t lds_m32(int l) { return l >= 16 ? 0x55555555u : (((1u << (2 * l)) - 1u) & 0x55555555u); }
// pos_verify for an entry that passed the pre-filter (bounds checked there), with the read's masks at hand: ml = its l
// nt, sd = its seed nt in the entry's orientation (both one bit per nt)
__device__ __forceinline__ bool pos_verify_fast(const RoundDev& P, const uint64_t* text, const uint64_t* bad, uint64_t q0, uint64_t q1,
                                                uint64_t ml0, uint64_t ml1, uint64_t sd0, uint64_t sd1, int l, uint32_t w, int& sm, int& tm) {
    const uint32_t wi = w >> 5; const int bs = (int)(w & 31) * 2;
    const uint64_t t0 = text[wi], t1 = text[wi + 1], t2 = text[wi + 2];
    const uint64_t x0 = bs ? (t0 >> bs) | (t1 << (64 - bs)) : t0, x1 = bs ? (t1 >> bs) | (t2 << (64 - bs)) : t1;
    uint64_t d0 = x0 ^ q0, d1 = x1 ^ q1;
    d0 = (d0 | (d0 >> 1)) & ml0; d1 = (d1 | (d1 >> 1)) & ml1;
    tm = popc64(d0) + popc64(d1);
    sm = popc64(d0 & sd0) + popc64(d1 & sd1);
    const uint32_t bi = w >> 6; const int sh = (int)(w & 63);
    const uint64_t b0 = bad[bi], b1 = bad[bi + 1];
    const uint64_t rb = sh ? (b0 >> sh) | (b1 << (64 - sh)) : b0;
    return tm <= P.maxmm && sm <= P.seedmm && (rb & ((1ull << l) - 1)) == 0;       // (l <= 56)
}
#ifndef MG_LDS_WAVES
#define MG_LDS_WAVES 4
#endif
__device__ __forceinline__ uint64_t uni64(uint64_t x) { return (uint64_t)uni((uint32_t)x) | ((uint64_t)uni((uint32_t)(x >> 32)) << 32); }
// rounds without block schemes (the exact-match round) need fewer registers: 12 waves per block, two blocks per CU
template <int NS> struct LdsCfg { static constexpr int BLOCK = 512, WAVES = BLOCK / 64, PER_SIMD = MG_LDS_WAVES; };
template <int RN, int NS>
__global__ void __launch_bounds__(LdsCfg<NS>::BLOCK, LdsCfg<NS>::PER_SIMD)
k_lds_round(const uint64_t* __restrict__ ukeys, uint32_t U, uint8_t* ann_round, int32_t* ann_ref, uint8_t* ann_amb,
            LibDev L, RoundDev P, RoundSch RS, unsigned long long* stats,
            ListIn in, ListOut out, uint32_t* fb_list, unsigned long long* fb_n, bool init,
            const unsigned long long* __restrict__ U_dev, uint32_t U0) {
    constexpr int BLK = LdsCfg<NS>::BLOCK, NWV = LdsCfg<NS>::WAVES;
    constexpr int NL = NS == 0 ? 3 : PROBE_MAX_LOOK;                // lookups a read can have: pieces / a scheme's shapes
    // LDS: [text | N + boundary bitmap | boundary bitmap, if the library holds N | boundaries before each bitmap word | waves]
    extern __shared__ uint64_t lds_dyn[];
    uint64_t* const s_text = lds_dyn;
    uint64_t* const s_bad = s_text + L.lds_tw;
    uint64_t* const s_sep = L.lds_sw ? s_bad + L.lds_bw : s_bad;
    uint32_t* const s_rank = reinterpret_cast<uint32_t*>(s_bad + L.lds_bw + L.lds_sw);
    LdsWave* const waves = reinterpret_cast<LdsWave*>(s_bad + L.lds_bw + L.lds_sw + (L.lds_bw + 1) / 2);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t j = tid; j < L.lds_tw; j += BLK) s_text[j] = L.text[j];
    for (uint32_t j = tid; j < L.lds_bw; j += BLK) { s_bad[j] = L.bad[j]; s_rank[j] = L.rank[j]; }
    for (uint32_t j = tid; j < L.lds_sw; j += BLK) s_sep[j] = L.sep[j];
    __syncthreads();
    if (U_dev) { const uint32_t ud = U0 + (uint32_t)(*U_dev >> 32); U = ud < U ? ud : U; }      // as in k_match_round; never beyond what the launch was sized for
    LdsWave& W = waves[tid >> 6];
    const uint32_t gw = blockIdx.x * NWV + (tid >> 6), nw = gridDim.x * NWV;
    const uint32_t ndense = (in.p && in.n) ? (uint32_t)*in.n : U;
    const uint32_t ntiles = in.K > 1 ? in.K * (in.cap / 64) : (ndense + 63) / 64;
    const unsigned long long below = (1ull << lane) - 1;
    // the reference that holds text position w: the boundaries (separators) before it, from the bitmap in LDS
    auto ref_at = [&](uint32_t w) -> uint32_t { return s_rank[w >> 6] + (uint32_t)__popcll(s_sep[w >> 6] & ones((int)(w & 63))); };
    // tile t of the launch's work: its list, that list's length, the tile's first element; false = nothing there
    auto tile_of = [&](uint32_t t, const uint32_t*& list, uint32_t& n, uint32_t& i0) -> bool {
        list = in.p;
        if (in.K > 1) {
            const uint32_t sk = t % in.K;
            n = (uint32_t)in.n[(size_t)sk * LIST_NSTRIDE]; i0 = (t / in.K) * 64; list += (size_t)sk * in.cap;
        } else { n = ndense; i0 = t * 64; }
        return i0 < n;
    };
    auto next_tile = [&](uint32_t t) -> uint32_t {                  // first tile >= t of this wave's stride that holds anything
        const uint32_t* l_; uint32_t n_, i_;
        while (t < ntiles && !tile_of(t, l_, n_, i_)) t += nw;
        return t;
    };
    uint32_t n_act = 0, n_hit = 0;                                  // wave totals, flushed once
    // the NEXT tile's unique ids and keys are fetched while this one is worked on
    uint32_t t = next_tile(gw);
    uint32_t u = EMPTY; uint64_t k[2] = {0, 0}; uint8_t done_before = 0; bool from_list = false;
    if (t < ntiles) {
        const uint32_t* list; uint32_t n, i0;
        tile_of(t, list, n, i0);
        const uint32_t i = i0 + lane;
        from_list = list != nullptr;
        u = i < n ? (list ? list[i] : i) : EMPTY;
        if (u != EMPTY) {
            if (!init && !list) done_before = ann_round[u];
            k[0] = ukeys[(uint64_t)u * 2]; k[1] = ukeys[(uint64_t)u * 2 + 1];
        }
    }
    while (t < ntiles) {
        const uint32_t tn = next_tile(t + nw);
        uint32_t un = EMPTY; bool nlist = false;
        if (tn < ntiles) {
            const uint32_t* list; uint32_t n, i0;
            tile_of(tn, list, n, i0);
            const uint32_t i = i0 + lane;
            nlist = list != nullptr;
            un = i < n ? (list ? list[i] : i) : EMPTY;
        }
        uint32_t* next_list = out.p;
        unsigned long long* next_n = out.n;
        uint32_t next_cap = 0xFFFFFFFFu;
        if (out.K > 1) {
            const uint32_t ok = t % out.K;
            next_list += (size_t)ok * out.cap; next_n += (size_t)ok * LIST_NSTRIDE; next_cap = out.cap;
        }
        if (u != EMPTY && init) { ann_round[u] = 0; ann_ref[u] = -1; ann_amb[u] = 0; }       // as in k_match_round
        const bool listed = u != EMPTY && done_before == 0;
        bool active = false;
        if (listed) {
            const int len = key_len<2>(k);
            active = !(P.len_lt && !(len < P.len_lt)) && !(P.len_gt && !(len > P.len_gt));
        }
        if (!active) { k[0] = 0; k[1] = 0; }
        ProbeRead R;
        probe_setup<NS>(k, L, P, RS, R);
        R.ok = R.ok && active;
        const bool run = R.ok && R.served;
        bool fb = R.ok && !R.served;
        W.best[0][lane] = 0; W.best[1][lane] = 0; W.best[2][lane] = 0;
        W.r0[lane] = R.r[0]; W.r1[lane] = R.r[1];
        W.ls[lane] = (uint32_t)(R.l & 255) | ((uint32_t)(R.s & 255) << 8);
        uint64_t c0 = 0, c1 = 0;
        revcomp2(R.r[0], R.r[1], R.l, c0, c1);
        W.cl[lane] = (uint32_t)c0;
        const int K = tab_K(R.tab, L);
        const bool ubits = tab_bits(R.tab);
        // every lookup of the read: occupancy words first, then the slots, all in flight at once
        uint32_t slot[NL], bb[NL]; int ps[NL]; bool have[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            have[j] = run && j < R.nlook;
            slot[j] = 0; bb[j] = 0; ps[j] = 0;
            if (have[j]) probe_lookup<NS == 1>(R, L, j, slot[j], bb[j], ps[j]);
        }
        uint32_t wd[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) wd[j] = (have[j] && ubits) ? L.sbits[bb[j] >> 5] : 0xFFFFFFFFu;
        // (the next tile's keys: its ids have been on their way since the top of the loop)
        uint64_t kn[2] = {0, 0}; uint8_t done_n = 0;
        if (un != EMPTY) {
            if (!init && !nlist) done_n = ann_round[un];
            kn[0] = ukeys[(uint64_t)un * 2]; kn[1] = ukeys[(uint64_t)un * 2 + 1];
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) have[j] = have[j] && ((wd[j] >> (bb[j] & 31)) & 1);
        CSlot sl[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) { sl[j] = CSlot{0, 0}; if (have[j]) sl[j] = L.cslots[slot[j]]; }
        wave_lds_sync();

        uint32_t b_sc = 0, b_kx = 0, b_kn = 0;                      // what this lane finds itself (single-entry buckets): sc32 / kx32 / kn32
        uint32_t sq_n = 0, lq_n = 0;                                // wave-uniform
        // one list entry against its owner's read (from LDS; the reverse complement is made here, for reverse entries)
        const uint32_t* const s_text32 = reinterpret_cast<const uint32_t*>(s_text);
        // an entry that passed the pre-filter, against its owner's read (from LDS; masks and reverse complement made here)
        auto record = [&](uint32_t owner, int sm, int tm, uint32_t w, int f) {
            atomicMax(&W.best[0][owner], sc32(sm, tm, w, f)); atomicMax(&W.best[1][owner], kx32(sm, w)); atomicMax(&W.best[2][owner], kn32(sm, w));
        };
        auto judge = [&](uint32_t e, uint32_t owner, int kk, int pss) {
            uint64_t q0 = W.r0[owner], q1 = W.r1[owner];
            const uint32_t ls = W.ls[owner];
            const int l = (int)(ls & 255), sd = (int)(ls >> 8);
            const bool rev = e >> 31;
            if (rev) { const uint64_t a0 = q0, a1 = q1; revcomp2(a0, a1, l, q0, q1); }
            const int sa = rev ? l - sd : 0, sb = rev ? l : sd;
            const uint32_t w = (e & POS_MASK) - (uint32_t)(rev ? l - pss - kk : pss);
            int sm, tm;
            if (pos_verify_fast(P, s_text, s_bad, q0, q1, ones(2 * l) & M55, ones(2 * l - 64) & M55, (ones(2 * sb) ^ ones(2 * sa)) & M55,
                                (ones(2 * sb - 64) ^ ones(2 * sa - 64)) & M55, l, w, sm, tm)) record(owner, sm, tm, w, rev ? 0 : 1);
        };
        // Both drains fetch the entries of several steps before they judge any: a list entry comes from the L2 or the
        // Infinity Cache (~1-2 us under load), and a drain that waited for each step's entries in turn ran at that pace.
        auto drain_short = [&]() {
            wave_lds_sync();
            constexpr int PER = 64 / LDS_SHORT;
            const uint32_t ql = lane / LDS_SHORT, ei = lane % LDS_SHORT;
            for (uint32_t base = 0; base < sq_n; base += PER * LDS_SB) {
                uint32_t e[LDS_SB];
#pragma unroll
                for (int j = 0; j < LDS_SB; ++j) {
                    const uint32_t q = base + j * PER + ql;
                    const uint64_t d = q < sq_n ? W.sq[q] : 0ull;
                    e[j] = ei < (uint32_t)((d >> 24) & 0xFFFF) ? L.cpos[(uint32_t)(d & 0xFFFFFF) + ei] : 0u;
                }
#pragma unroll 1
                for (int j = 0; j < LDS_SB; ++j) {
                    const uint32_t q = base + j * PER + ql;
                    const uint32_t e0 = e[0];
#pragma unroll
                    for (int x = 0; x + 1 < LDS_SB; ++x) e[x] = e[x + 1];
                    if (base + j * PER >= sq_n) break;
                    const uint64_t d = q < sq_n ? W.sq[q] : 0ull;
                    if (ei < (uint32_t)((d >> 24) & 0xFFFF)) {
                        const uint32_t owner = (uint32_t)(d >> 40) & 63; const int kk = (int)(d >> 52) & 63, pss = (int)(d >> 46) & 63;
                        const int l = (int)(W.ls[owner] & 255);
                        if (lds_prefilter(s_text32, L.T, e0, l, pss, l - pss - kk, (uint32_t)W.r0[owner], W.cl[owner], lds_m32(l), P.maxmm)) judge(e0, owner, kk, pss);
                    }
                }
            }
            sq_n = 0;
            wave_lds_sync();
        };
        auto drain_long = [&]() {
            wave_lds_sync();
            for (uint32_t q0 = 0; q0 < lq_n; q0 += LDS_LB) {
                uint32_t e[LDS_LB];
#pragma unroll
                for (int j = 0; j < LDS_LB; ++j) {
                    const uint64_t d = W.lq[q0 + j < lq_n ? q0 + j : lq_n - 1];
                    const uint32_t lo = uni((uint32_t)(d & 0xFFFFFF)), cnt = q0 + j < lq_n ? uni((uint32_t)((d >> 24) & 0xFFFF)) : 0u;
                    e[j] = lane < cnt ? L.cpos[lo + lane] : 0u;
                }
#pragma unroll 1
                for (int j = 0; j < LDS_LB; ++j) {
                    uint32_t e0 = e[0];
#pragma unroll
                    for (int x = 0; x + 1 < LDS_LB; ++x) e[x] = e[x + 1];
                    if (q0 + j >= lq_n) break;
                    const uint64_t d = W.lq[q0 + j];
                    const uint32_t lo = uni((uint32_t)(d & 0xFFFFFF)), cnt = uni((uint32_t)((d >> 24) & 0xFFFF));
                    const uint32_t meta = uni((uint32_t)(d >> 40));      // owner | ps << 6 | K << 12
                    // ONE owner: its read's pre-filter words sit in scalar registers
                    const uint32_t owner = meta & 63; const int kk = (int)(meta >> 12) & 63, pss = (int)(meta >> 6) & 63;
                    const uint32_t ls = uni(W.ls[owner]);
                    const int l = (int)(ls & 255), sd = (int)(ls >> 8);
                    const uint64_t o0 = uni64(W.r0[owner]), o1 = uni64(W.r1[owner]);
                    uint64_t oc0, oc1;
                    revcomp2(o0, o1, l, oc0, oc1);
                    const uint64_t oml0 = ones(2 * l) & M55, oml1 = ones(2 * l - 64) & M55;
                    const uint64_t osf0 = ones(2 * sd) & M55, osf1 = ones(2 * sd - 64) & M55;
                    const uint64_t osr0 = oml0 & ~ones(2 * (l - sd)), osr1 = oml1 & ~ones(2 * (l - sd) - 64);
                    const uint32_t q_f = (uint32_t)o0, q_r = (uint32_t)oc0, m32 = lds_m32(l);
                    const int off_r = l - pss - kk;
                    for (uint32_t c = 0; c < cnt; c += 64) {
                        if (c) e0 = c + lane < cnt ? L.cpos[lo + c + lane] : 0u;          // (a list of more than 64 entries: the further strides, one load each)
                        if (c + lane < cnt && lds_prefilter(s_text32, L.T, e0, l, pss, off_r, q_f, q_r, m32, P.maxmm)) {
                            const bool rev = e0 >> 31;
                            const uint32_t w = (e0 & POS_MASK) - (uint32_t)(rev ? off_r : pss);
                            int sm, tm;
                            if (pos_verify_fast(P, s_text, s_bad, rev ? oc0 : o0, rev ? oc1 : o1, oml0, oml1, rev ? osr0 : osf0, rev ? osr1 : osf1, l, w, sm, tm))
                                record(owner, sm, tm, w, rev ? 0 : 1);
                        }
                    }
                }
            }
            lq_n = 0;
            wave_lds_sync();
        };
        // the read's masks in either orientation, for the entries this lane checks itself
        const uint64_t mr0 = R.ml[0] & ~ones(2 * (R.l - R.s)), mr1 = R.ml[1] & ~ones(2 * (R.l - R.s) - 64);
        const uint32_t m32 = lds_m32(R.l);
        // one lookup per iteration, element 0 of the register arrays (which are shifted down each time: the loop is not
        // unrolled, so both drains are inlined once); a last iteration without a lookup empties the queues
#pragma unroll 1
        for (int j = 0; j <= NL; ++j) {
            const bool last = j == NL;
            const bool hv = !last && have[0];
            const CSlot s0 = sl[0];
            const int p0 = ps[0];
#pragma unroll
            for (int q = 0; q + 1 < NL; ++q) { have[q] = have[q + 1]; sl[q] = sl[q + 1]; ps[q] = ps[q + 1]; }
            have[NL - 1] = false;
            if (!last && !__any(hv)) continue;
            // lookups that what is known by now rules out are dropped (own finds + what the drains so far brought)
            const uint32_t known = b_sc > W.best[0][lane] ? b_sc : W.best[0][lane];
            const bool need = hv && !fb && probe_needed(R, P, j, sc32_as_score(known));
            bool is_s = false, is_l = false;
            uint32_t cnt = 0;
            if (need) {
                if (s0.b == CS_SINGLE) {
                    const bool rev = s0.a >> 31;
                    const int off = rev ? R.l - p0 - K : p0;
                    if (lds_prefilter(s_text32, L.T, s0.a, R.l, off, off, (uint32_t)R.r[0], (uint32_t)c0, m32, P.maxmm)) {
                        const uint32_t w = (s0.a & POS_MASK) - (uint32_t)off;
                        int sm, tm;
                        if (pos_verify_fast(P, s_text, s_bad, rev ? c0 : R.r[0], rev ? c1 : R.r[1], R.ml[0], R.ml[1], rev ? mr0 : R.ms[0], rev ? mr1 : R.ms[1], R.l, w, sm, tm))
                            { const uint32_t x = sc32(sm, tm, w, rev ? 0 : 1), y = kx32(sm, w), z = kn32(sm, w);
                              b_sc = x > b_sc ? x : b_sc; b_kx = y > b_kx ? y : b_kx; b_kn = z > b_kn ? z : b_kn; }
                    }
                } else {
                    cnt = s0.b - s0.a;
                    if (cnt > (uint32_t)LDS_MAX_LIST) fb = true;
                    else if (cnt > (uint32_t)LDS_SHORT) is_l = true;
                    else if (cnt) is_s = true;
                }
            }
            const unsigned long long ms = __ballot(is_s), ml = __ballot(is_l);
            if (sq_n && (last || sq_n + (uint32_t)__popcll(ms) > (uint32_t)LDS_SQ)) drain_short();
            if (is_s) W.sq[sq_n + (uint32_t)__popcll(ms & below)] = lds_desc(s0.a, cnt, lane, p0, K);
            sq_n += (uint32_t)__popcll(ms);
            if (lq_n && (last || lq_n + (uint32_t)__popcll(ml) > (uint32_t)LDS_LQ)) drain_long();
            if (is_l) W.lq[lq_n + (uint32_t)__popcll(ml & below)] = lds_desc(s0.a, cnt, lane, p0, K);
            lq_n += (uint32_t)__popcll(ml);
        }
        wave_lds_sync();

        // ---- results: the reference of a position is a rank query on the boundary bitmap in LDS
        bool hit = false;
        if (run && !fb) {
            const uint32_t a = W.best[0][lane], b2 = W.best[1][lane], c2 = W.best[2][lane];
            b_sc = a > b_sc ? a : b_sc; b_kx = b2 > b_kx ? b2 : b_kx; b_kn = c2 > b_kn ? c2 : b_kn;
            if (b_sc) {
                hit = true;
                const uint32_t r = ref_at((b_sc >> 1) & 0xFFFFFFu);
                const bool amb = ref_at(b_kx & 0xFFFFFFu) != ref_at(0xFFFFFFu - (b_kn & 0xFFFFFFu));
                ann_round[u] = (uint8_t)(RN + 1); ann_ref[u] = (int32_t)r; ann_amb[u] = (uint8_t)amb;
            }
        }
        n_act += (uint32_t)__popcll(__ballot(active && !fb)); n_hit += (uint32_t)__popcll(__ballot(hit));
        const bool surv = next_list != nullptr && listed && !hit && !fb;
        const unsigned long long sv = __ballot(surv), fv = __ballot(fb);
        if (sv) {
            uint32_t base = 0;
            if (lane == 0) base = (uint32_t)atomicAdd(next_n, (unsigned long long)__popcll(sv));
            base = uni(base);
            const uint32_t at = base + (uint32_t)__popcll(sv & below);
            if (surv && at < next_cap) next_list[at] = u;       // (cannot fail: cap = the tiles that share the sub-list x 64)
        }
        if (fv) {
            uint32_t base = 0;
            if (lane == 0) base = (uint32_t)atomicAdd(fb_n, (unsigned long long)__popcll(fv));
            base = uni(base);
            if (fb) fb_list[base + (uint32_t)__popcll(fv & below)] = u;
        }
        wave_lds_sync();                                           // the wave's LDS area is rewritten by its next tile
        t = tn; u = un; k[0] = kn[0]; k[1] = kn[1]; done_before = done_n; from_list = nlist;
    }
    (void)from_list;
    if (lane == 0) {
        const uint32_t slot_ = gw & (STAT_SLOTS - 1);
        if (n_act) atomicAdd(&stats[(size_t)slot_ * 8], (unsigned long long)n_act);
        if (n_hit) atomicAdd(&stats[(size_t)(STAT_SLOTS + slot_) * 8], (unsigned long long)n_hit);
    }
}

// ------------------------------------------------------------------------------------------
// count vectors (miRge.pl:465-493) over one sample's (unique id, count) entries.
// Per-block partial sums live in LDS (category totals, length histogram, and -- when the
// miRNA library is small enough -- the whole quant/iscan column), then one flush of atomics.
constexpr int CNT_LDS_M = 3584;        // 2 x 3584 x 8 B = 56 KB of dynamic LDS (below the 64 KB a launch may ask for without a function attribute)
// dynamic LDS: 2 * M words when M <= CNT_LDS_M (the quant and iscan columns), else none -- a human-sized miRNA library
// (2.7 k names) takes 43 KB and three blocks fit a CU, where a fixed 6144-name array held one
template <int KW>
__global__ void __launch_bounds__(1024)
k_counts(const uint32_t* __restrict__ coo_g, const uint32_t* __restrict__ coo_c, uint32_t n, int sample, int S,
         const uint64_t* __restrict__ ukeys, const uint8_t* __restrict__ ann_round, const int32_t* __restrict__ ann_ref,
         int M, uint32_t round_libs, uint32_t iscan_rounds, unsigned long long* quant, unsigned long long* iscan,
         unsigned long long* cat, unsigned long long* lenhist, unsigned long long* tuniq,
         const unsigned long long* __restrict__ n_dev, bool lds_cols) {
    // lds_cols: the launch brought 2 * M words of dynamic LDS for the quant / iscan columns (the host leaves them out
    // for a handful of entries: a 43 KB block finds no room on CUs that the cascade's blocks fill, and the 48-byte
    // class's 40 K entries waited 0.45 ms for one)
    extern __shared__ unsigned long long s_dyn[];
    if (n_dev) { const uint32_t nd = (uint32_t)(*n_dev & 0xFFFFFFFFull); n = nd < n ? nd : n; }      // the entry count is still on the device (low half of the collapse's rank counter); n was an upper bound
    __shared__ unsigned long long s_cat[5];
    __shared__ unsigned long long s_len[128];
    __shared__ unsigned long long s_tu;
    const bool ldsm = M <= CNT_LDS_M && M > 0 && (int)n >= 0 && lds_cols;
    unsigned long long* s_q = s_dyn;
    unsigned long long* s_i = s_dyn + (ldsm ? M : 0);
    for (int j = threadIdx.x; j < 128; j += blockDim.x) s_len[j] = 0;
    if (threadIdx.x < 5) s_cat[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_tu = 0;
    if (ldsm) for (int j = threadIdx.x; j < M; j += blockDim.x) { s_q[j] = 0; s_i[j] = 0; }
    __syncthreads();
    // the five category totals and the entry count are kept per thread and folded once: as LDS atomics every lane of
    // every wave hit the same six words
    unsigned long long c_cat[5] = {0, 0, 0, 0, 0}, c_tu = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        uint32_t g = coo_g[e];
        unsigned long long c = coo_c[e];
        int len = key_len<KW>(&ukeys[(uint64_t)g * KW]);
        int r = ann_round[g];
        c_tu += 1;                                                   // miRge.pl:468
        atomicAdd(&s_len[len & 127], c);                             // miRge.pl:315
        // round_libs: 2 bits per round = the library it ran against; the read's category is that library's (for the
        // reference cascade: rounds 1 and 5 -> miRNA, 2 -> hairpin, 3 -> other, 4 -> mRNA), and only a round against the
        // miRNA library indexes quant / iscan -- whatever cascade mirge_set_rounds installed
        int catidx = 4;                                              // miRge.pl:489 remReads
        if (r > 0) {
            const int lib = (int)((round_libs >> (2 * (r - 1))) & 3u);
            catidx = lib;                                            // :469 miRNA, :480 hairpin, :483 other, :486 mRNA
            if (lib == 0) {
                const int32_t m = ann_ref[g];
                const bool isc = (iscan_rounds >> (r - 1)) & 1u;     // miRge.pl:474: exact-miRNA round only
                if (ldsm) { atomicAdd(&s_q[m], c); if (isc) atomicAdd(&s_i[m], c); }
                else {
                    atomicAdd(&quant[(uint64_t)m * S + sample], c);  // miRge.pl:473 / 476
                    if (isc) atomicAdd(&iscan[(uint64_t)m * S + sample], c);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) c_cat[q] += (q == catidx) ? c : 0ull;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        unsigned long long v = c_cat[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cat[q], v);
    }
    {
        unsigned long long v = c_tu;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_tu, v);
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


// ------------------------------------------------------------------------------------------
// One cascade round over the verbatim uniques (match_long), and their share of the count vectors.  A rare class:
// thread per unique, global atomics.
__global__ void __launch_bounds__(256)
k_match_verb(const uint64_t* __restrict__ pool, const uint64_t* __restrict__ off, const uint32_t* __restrict__ len, uint32_t U,
             uint8_t* ann_round, int32_t* ann_ref, uint8_t* ann_amb, LibDev L, RoundDev P, int r, unsigned long long* stats) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = false, hit = false;
    if (u < U) {
        uint8_t before = 0;
        if (r == 0) { ann_round[u] = 0; ann_ref[u] = -1; ann_amb[u] = 0; }
        else before = ann_round[u];
        const int l = (int)len[u];
        active = before == 0 && !(P.len_lt && !(l < P.len_lt)) && !(P.len_gt && !(l > P.len_gt));      // miRge.pl:382-393
        if (active) {
            uint64_t score, kmax, kmin;
            match_long(pool + off[u], l, L, P, score, kmax, kmin);
            if (score) {
                hit = true;
                const uint32_t ref = ref_of(L, (uint32_t)(score >> 1));
                ann_round[u] = (uint8_t)(r + 1); ann_ref[u] = (int32_t)ref;
                ann_amb[u] = (uint8_t)(ref_of(L, (uint32_t)kmax) != ref_of(L, 0xFFFFFFFFu - (uint32_t)kmin));
            }
        }
    }
    const unsigned long long am = __ballot(active), hm = __ballot(hit);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) & (STAT_SLOTS - 1);
        if (am) atomicAdd(&stats[(size_t)slot * 8], (unsigned long long)__popcll(am));
        if (hm) atomicAdd(&stats[(size_t)(STAT_SLOTS + slot) * 8], (unsigned long long)__popcll(hm));
    }
}
// entries (unique, sample, count), at most one per pair
__global__ void __launch_bounds__(256)
k_counts_verb(const uint32_t* __restrict__ coo_u, const uint32_t* __restrict__ coo_s, const uint32_t* __restrict__ coo_c, uint32_t n, int S,
              const uint32_t* __restrict__ len, const uint8_t* __restrict__ ann_round, const int32_t* __restrict__ ann_ref,
              uint32_t round_libs, uint32_t iscan_rounds, int len_bins, unsigned long long* quant, unsigned long long* iscan,
              unsigned long long* cat, unsigned long long* lenhist, unsigned long long* tuniq) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint32_t g = coo_u[e], s = coo_s[e];
    const unsigned long long c = coo_c[e];
    const int r = ann_round[g];
    atomicAdd(&tuniq[s], 1ull);                                                  // miRge.pl:468
    if ((int)len[g] < len_bins) atomicAdd(&lenhist[(uint64_t)len[g] * S + s], c);  // miRge.pl:315
    int catidx = 4;
    if (r > 0) {
        const int lib = (int)((round_libs >> (2 * (r - 1))) & 3u);
        catidx = lib;
        if (lib == 0) {
            const int32_t m = ann_ref[g];
            atomicAdd(&quant[(uint64_t)m * S + s], c);
            if ((iscan_rounds >> (r - 1)) & 1u) atomicAdd(&iscan[(uint64_t)m * S + s], c);
        }
    }
    atomicAdd(&cat[(uint64_t)catidx * S + s], c);
}

}  // namespace mg
