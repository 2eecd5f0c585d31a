// trim.h -- read trimming on the device: raw FASTQ records -> trimmed 2-bit keys.
//
// Replaces, per record, what the reference runs before quantReads (miRge.pl:259-261, 268-289):
//   trim_file.py:32      QualityTrimmer(0, 10, base)              BWA-style 3' quality trimming
//   trim_file.py:36-43   AdapterCutter(3' adapters, e = 0.12) | UnconditionalCutter(+N) | nothing ("none")
//   trim_file.py:57      keep if len >= 16
//   miRge.pl:261         cutadapt -u trim5 -u -trim3              (no length filter)
// PARITY UNPINNED: the algorithms are cutadapt's (third party, absent from the reference, no test vector there); they
// are restated from its 1.9-1.11 sources -- see oracle/trim_oracle.c, which the tests compare this file with.
//   quality_trim_index : s accumulates (cutoff - q) from the 3' end, stop = position of the maximum, break at s < 0
//   Adapter.match_to   : leftmost exact occurrence of the adapter, else Aligner.locate: edit-distance DP of the adapter
//                        (rows) against the read (columns), alignment free to start anywhere in the read and to end inside
//                        the adapter at the read's end, Ukkonen band (rows whose cost exceeds k = int(rate * m) are
//                        dropped), best = most matches then lowest cost among cells with cost <= rows * rate, overlap >= 3
// One lane per record; the DP column (<= 33 cells of cost | matches | origin) lives in LDS, one word per cell and lane.
#pragma once
#include "fastq.h"
#include <cstdlib>
#include <cstring>
#include <string>

namespace mg {

constexpr int TRIM_MAX_ADAPTERS = 16, TRIM_MAX_ALEN = 64;

struct TrimDev {
    int32_t mode;                          // 0 = no adapter step, 1 = 3' adapters, 2 = unconditional cut
    int32_t uncond;                        // mode 2: > 0 removes first bases, < 0 last bases
    int32_t n_ad;
    int32_t alen[TRIM_MAX_ADAPTERS];
    uint32_t anchored;                     // bit ad: the adapter was written "ADAPTER$" (must end at the read's 3' end)
    int32_t max_alen;
    uint64_t acode[TRIM_MAX_ADAPTERS][2];  // 2-bit codes, adapter nt i at bits 2i of the 128-bit value (plain adapters: the exact-occurrence search)
    uint64_t amask[TRIM_MAX_ADAPTERS][4];  // 4 bits per adapter nt: the bases it stands for (A = bit 0 .. T = bit 3; IUPAC codes several, N all, X none)
    uint32_t wild;                         // bit ad: the adapter holds a character outside ACGT (cutadapt's adapter_wildcards are then in force)
    int32_t base;                          // phred base of the quality trimmer
    int32_t trim5, trim3, min_len, min_overlap;
    int8_t  maxerr[TRIM_MAX_ALEN + 1];     // [len] = largest cost with (double)cost <= len * rate   (host, in double)
    int8_t  maxerr2[TRIM_MAX_ALEN + 1];    // [len] = largest cost with !((double)cost / len > rate) (host, in double)
};

// --adapter as miRge.pl takes it -> TrimDev (host).  0 ok, -1 bad argument, -5 over a limit; msg says what.
inline int trim_dev_from_spec(const char* adapter, int trim5, int trim3, TrimDev& T, std::string& msg) {
    memset(&T, 0, sizeof T);
    std::string a = adapter ? adapter : "none";
    if (a == "illumina") a = "TGGAATTCTCGGGTGCCAAGGAACTCCAG";       // miRge.pl:81-83
    else if (a == "ion") a = "+11";                                  // miRge.pl:84-86
    if (trim5 < 0 || trim3 < 0) { msg = "trim5 / trim3 must not be negative"; return -1; }
    T.trim5 = trim5; T.trim3 = trim3; T.min_len = 16; T.min_overlap = 3; T.base = 33;
    const double rate = 0.12;                                        // trim_file.py:34
    for (int len = 0; len <= TRIM_MAX_ALEN; ++len) {
        int e = 0; while ((double)(e + 1) <= len * rate) ++e;
        T.maxerr[len] = (int8_t)e;
        int e2 = 0; while (len > 0 && !((double)(e2 + 1) / len > rate)) ++e2;
        T.maxerr2[len] = (int8_t)e2;
    }
    if (a == "none" || a.empty()) { T.mode = 0; return 0; }
    if (a[0] == '+' || a[0] == '-') { T.mode = 2; T.uncond = atoi(a.c_str()); return 0; }
    T.mode = 1;
    size_t p = 0;
    while (p <= a.size()) {
        size_t e = a.find(',', p); if (e == std::string::npos) e = a.size();
        std::string one = a.substr(p, e - p);
        p = e + 1;
        // cutadapt's gather_adapters (trim_file.py:17): "name=SEQUENCE"; a trailing "$" anchors a 3' adapter at the read's end.
        // Adapter.__init__: upper case, U -> T.
        const size_t eq = one.find('=');
        if (eq != std::string::npos) one = one.substr(eq + 1);
        bool anchored = !one.empty() && one.back() == '$';
        if (anchored) one.pop_back();
        if (one.empty()) continue;
        if (T.n_ad >= TRIM_MAX_ADAPTERS) { msg = "more than " + std::to_string(TRIM_MAX_ADAPTERS) + " adapters"; return -5; }
        if ((int)one.size() > TRIM_MAX_ALEN) { msg = "adapter longer than " + std::to_string(TRIM_MAX_ALEN) + " nt"; return -5; }
        if (one.find("...") != std::string::npos) { msg = "adapter '" + one + "': linked adapters (ADAPTER1...ADAPTER2, cutadapt >= 1.10) are not supported"; return -1; }
        for (size_t i = 0; i < one.size(); ++i) {
            uint8_t ch = (uint8_t)one[i];
            if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);
            if (ch == 'U') ch = 'T';
            // the IUPAC table of cutadapt's aligner: a bit per base the character stands for
            static const char* codes = "ACGTRYSWKMBDHVNX";
            static const uint8_t bits[16] = {1, 2, 4, 8, 1 | 4, 2 | 8, 2 | 4, 1 | 8, 4 | 8, 1 | 2, 2 | 4 | 8, 1 | 4 | 8, 1 | 2 | 8, 1 | 2 | 4, 15, 0};
            const char* at = ch ? strchr(codes, (int)ch) : nullptr;
            if (!at) { msg = "adapter '" + one + "': 3' adapters of A, C, G, T and IUPAC wildcard codes (plain, or anchored with a trailing $) are supported; braces, 5' and linked adapters are not"; return -1; }
            const int ci = (int)(at - codes);
            if (ci < 4) T.acode[T.n_ad][i >> 5] |= (uint64_t)ci << (2 * (i & 31));
            else T.wild |= 1u << T.n_ad;
            T.amask[T.n_ad][i >> 4] |= (uint64_t)bits[ci] << (4 * (i & 15));
        }
        if (anchored) T.anchored |= 1u << T.n_ad;
        T.alen[T.n_ad] = (int32_t)one.size(); T.max_alen = std::max<int32_t>(T.max_alen, (int32_t)one.size()); ++T.n_ad;
    }
    if (!T.n_ad) T.mode = 0;
    return 0;
}

MG_HD int code_at(const uint64_t (&sq)[4], int j) { return (int)((pick<4>(sq, j >> 5) >> (2 * (j & 31))) & 3); }
MG_HD bool isn_at(const uint64_t (&nm)[2], int j) { return ((j < 64 ? nm[0] : nm[1]) >> (j & 63)) & 1; }
// How the adapter aligner sees a read: a 2-bit code and "matches nothing" per position.  PackedAcc: a read held as key
// words (<= 124 nt of A C G T N).  ByteAcc: the sequence line where it lies in the file -- lines no key holds (longer,
// or with other characters: lower case counts as its base, as cutadapt compares the upper-cased read; anything else
// matches no adapter base, as N does).
struct PackedAcc {
    const uint64_t (&sq)[4]; const uint64_t (&nm)[2];
    MG_HD int code(int j) const { return code_at(sq, j); }
    MG_HD bool isn(int j) const { return isn_at(nm, j); }
};
struct ByteAcc {
    const uint8_t* __restrict__ line;
    MG_HD static int up(uint8_t ch) { return base_code((ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch); }      // match_to compares read.sequence.upper()
    MG_HD int code(int j) const { const int c = up(line[j]); return c < 0 || c > 3 ? 0 : c; }
    MG_HD bool isn(int j) const { const int c = up(line[j]); return c < 0 || c > 3; }
};

// DP cell: cost (bits 24-31) | matches (16-23) | origin (0-15)
MG_HD uint32_t cell(int cost, int matches, int origin) { return ((uint32_t)cost << 24) | ((uint32_t)matches << 16) | (uint32_t)origin; }
MG_HD int c_cost(uint32_t c) { return (int)(c >> 24); }
MG_HD int c_matches(uint32_t c) { return (int)((c >> 16) & 255); }
MG_HD int c_origin(uint32_t c) { return (int)(c & 0xFFFF); }

// Adapter `ad` against read nts [a, b): true if it matches; rstart (relative to a) and matches out.
// col: this lane's DP column, cell i at col[i * stride].
MG_HD int ad_code(const TrimDev& T, int ad, int i) { return (int)((T.acode[ad][i >> 5] >> (2 * (i & 31))) & 3); }
// does adapter nt i stand for the read base of code c (cutadapt: the two characters' bit sets meet)
MG_HD bool ad_eq(const TrimDev& T, int ad, int i, int c) { return (T.amask[ad][i >> 4] >> (4 * (i & 15) + c)) & 1; }
template <class Acc>
MG_HD bool adapter_match(const TrimDev& T, int ad, const Acc& rd, int a, int b,
                         uint32_t* col, int stride, int& rstart, int& matches_out) {
    const int m = T.alen[ad], n = b - a;
    const bool anch = (T.anchored >> ad) & 1u;
    // exact occurrence first: the leftmost one (anchored: only as the read's suffix) -- unless the adapter holds
    // wildcards (Adapter.match_to: "try to find an exact match first unless wildcards are allowed")
    if ((T.wild >> ad) & 1u) {
    } else if (m <= 32 && !anch) {
        // rolling 2m-bit window over the read
        const uint64_t ac = T.acode[ad][0];
        const uint64_t full = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1);
        uint64_t h = 0; int nrun = 0;                        // nrun: nts since the last N
        for (int j = 0; j < n; ++j) {
            const int c = rd.code(a + j);
            const bool isn = rd.isn(a + j);
            h = ((h >> 2) | ((uint64_t)c << (2 * (m - 1)))) & full;
            nrun = isn ? 0 : nrun + 1;
            if (j >= m - 1 && nrun >= m && h == ac) { rstart = j - m + 1; matches_out = m; return true; }
        }
    } else {
        for (int p = anch ? (n - m > 0 ? n - m : 0) : 0; p + m <= n; ++p) {
            int i = 0;
            while (i < m && !rd.isn(a + p + i) && rd.code(a + p + i) == ad_code(T, ad, i)) ++i;
            if (i == m) { rstart = p; matches_out = m; return true; }
        }
    }
    const int k = T.maxerr[m];
    for (int i = 0; i <= m; ++i) col[i * stride] = cell(i, 0, 0);
    int best_matches = 0, best_cost = m + n, best_origin = 0, best_ref_stop = m;
    int last = m < k + 1 ? m : k + 1;
    bool done = false;
    for (int j = 1; j <= n && !done; ++j) {
        const int cj = rd.code(a + j - 1);
        const bool nj = rd.isn(a + j - 1);
        uint32_t tmp = col[0];
        uint32_t up = cell(0, 0, j);                          // column[0]: cost 0, origin j
        col[0] = up;
        for (int i = 1; i <= last; ++i) {
            const uint32_t cur = col[i * stride];
            const bool eq = !nj && ad_eq(T, ad, i - 1, cj);
            uint32_t nw;
            if (eq) nw = cell(c_cost(tmp), c_matches(tmp) + 1, c_origin(tmp));
            else {
                const int cd = c_cost(tmp) + 1, cdel = c_cost(cur) + 1, cins = c_cost(up) + 1;
                if (cd <= cdel && cd <= cins) nw = cell(cd, c_matches(tmp), c_origin(tmp));
                else if (cins <= cdel) nw = cell(cins, c_matches(up), c_origin(up));
                else nw = cell(cdel, c_matches(cur), c_origin(cur));
            }
            tmp = cur; up = nw;
            col[i * stride] = nw;
        }
        while (last >= 0 && c_cost(col[last * stride]) > k) --last;
        if (last < m) ++last;
        else if (!anch) {                                     // an anchored adapter may only end where the read ends
            const uint32_t c = col[m * stride];
            const int cost = c_cost(c), matches = c_matches(c);
            if (m >= T.min_overlap && cost <= T.maxerr[m] && (matches > best_matches || (matches == best_matches && cost < best_cost))) {
                best_matches = matches; best_cost = cost; best_origin = c_origin(c); best_ref_stop = m;
                if (cost == 0 && matches == m) done = true;
            }
        }
    }
    for (int i = m; i >= (anch ? m : 0); --i) {               // ... and must be taken in whole
        const uint32_t c = col[i * stride];
        const int cost = c_cost(c), matches = c_matches(c);
        if (i >= T.min_overlap && cost <= T.maxerr[i] && (matches > best_matches || (matches == best_matches && cost < best_cost))) {
            best_matches = matches; best_cost = cost; best_origin = c_origin(c); best_ref_stop = i;
        }
    }
    if (best_cost == m + n) return false;
    const int length = best_ref_stop;
    if (length < T.min_overlap || best_cost > T.maxerr2[length]) return false;
    rstart = best_origin; matches_out = best_matches;
    return true;
}

// quality line [qs, qs + n) -> [start, stop) as cutadapt's quality_trim_index(q, 0, 10, base)
MG_HD void quality_trim(const uint8_t* __restrict__ data, uint64_t qs, int n, int base, int& start, int& stop) {
    const uint64_t* d64 = reinterpret_cast<const uint64_t*>(data);
    uint64_t w = 0, wi = ~0ull;
    int s = 0, mx = 0;
    start = 0; stop = n;
    for (int i = 0; i < n; ++i) {
        const uint64_t p = qs + (uint64_t)i;
        if ((p >> 3) != wi) { wi = p >> 3; w = d64[wi]; }
        s += 0 - ((int)((w >> (8 * (p & 7))) & 255) - base);
        if (s < 0) break;
        if (s > mx) { mx = s; start = i + 1; }
    }
    s = 0; mx = 0;
    for (int i = n - 1; i >= 0; --i) {
        const uint64_t p = qs + (uint64_t)i;
        if ((p >> 3) != wi) { wi = p >> 3; w = d64[wi]; }
        s += 10 - ((int)((w >> (8 * (p & 7))) & 255) - base);
        if (s < 0) break;
        if (s > mx) { mx = s; stop = i; }
    }
    if (start >= stop) { start = 0; stop = 0; }
}

// the trimming steps on read positions [a, b) after the quality trimmer: adapter or unconditional cut, the length
// filter, then trim5 / trim3.  false = dropped.
template <class Acc>
MG_HD bool trim_steps(const TrimDev& T, const Acc& rd, uint32_t* col, int stride, int& a, int& b) {
    if (T.mode == 2) {
        if (T.uncond > 0) a = a + T.uncond < b ? a + T.uncond : b;
        else if (T.uncond < 0) b = b + T.uncond > a ? b + T.uncond : a;
    } else if (T.mode == 1) {
        int best_matches = -1, best_rstart = -1;
        for (int ad = 0; ad < T.n_ad; ++ad) {
            int rs = 0, mt = 0;
            if (adapter_match(T, ad, rd, a, b, col, stride, rs, mt) && mt > best_matches) { best_matches = mt; best_rstart = rs; }
        }
        if (best_rstart >= 0) b = a + best_rstart;
    }
    if (b - a < T.min_len) return false;
    if (T.trim5 > 0) a = a + T.trim5 < b ? a + T.trim5 : b;
    if (T.trim3 > 0) b = b - T.trim3 > a ? b - T.trim3 : a;
    return true;
}

// One record.  s / qs / qe: offsets of its sequence line, quality line and the end of the quality line.
// returns 0 = dropped (too short), 1 = kept as a key (key words + len out), 2 = kept as a verbatim read (what is left
// of it, [oa, ob) of the sequence line, fits no key), or an error code: 4 control character, 5 line longer than
// FQ_MAX_LINE, 6 sequence and quality lines differ in length.
MG_HD int trim_record(const uint8_t* __restrict__ data, uint64_t size, uint64_t s, uint64_t qs, uint64_t qe, const TrimDev& T,
                      uint32_t* col, int stride, uint64_t (&ks)[4], uint64_t (&kn)[2], int& klen, int& oa, int& ob) {
    uint64_t sq[4], nm[2]; int len; bool hasN;
    int rc = fq_pack_read<true>(data, size, s, sq, nm, len, hasN);
    if (qe > qs && data[qe - 1] == '\r') --qe;                 // cutadapt's FastqReader strips line ends
    int a, b;
    if (rc == 0) {
        if ((int)(qe - qs) != len) return 6;
        quality_trim(data, qs, len, T.base, a, b);
        if (!trim_steps(T, PackedAcc{sq, nm}, col, stride, a, b)) return 0;
        klen = b - a;
        shr<4>(sq, 2 * a, ks); keep_low<4>(ks, 2 * klen);
        shr<2>(nm, a, kn); keep_low<2>(kn, klen);
        return 1;
    }
    // the raw line fits no key (150-nt reads with the adapter still on, lower case, IUPAC codes ...): the same steps on
    // its bytes; what is left is packed if it fits a key by then
    rc = fq_line_scan<true>(data, size, s, len);
    if (rc) return rc;
    if ((int)(qe - qs) != len) return 6;
    quality_trim(data, qs, len, T.base, a, b);
    if (!trim_steps(T, ByteAcc{data + s}, col, stride, a, b)) return 0;
    klen = b - a;
    bool fits = klen <= 124;
    for (int j = a; j < b && fits; ++j) fits = base_code(data[s + j]) >= 0;
    if (!fits) { oa = a; ob = b; return 2; }
    ks[0] = ks[1] = ks[2] = ks[3] = 0; kn[0] = kn[1] = 0;
    for (int j = 0; j < klen; ++j) {
        const int c = base_code(data[s + a + j]);
        const uint64_t cc = c < 4 ? (uint64_t)c : 0ull;
#pragma unroll
        for (int q = 0; q < 4; ++q) ks[q] |= (q == (j >> 5)) ? cc << (2 * (j & 31)) : 0ull;
        if (c == 4) { if (j < 64) kn[0] |= 1ull << j; else kn[1] |= 1ull << (j - 64); }
    }
    return 1;
}

#ifdef __HIPCC__
constexpr int TRIM_BLOCK = 256;            // 128 when an adapter is longer than 32 nt: the DP columns of a block are LDS, (max_alen + 1) words per lane
// counters: [0] n16, [1] n48, [2] error code, [3] first offending record, [4] records kept, [7] verbatim reads, [8] their bytes
__global__ void __launch_bounds__(TRIM_BLOCK)
k_fq_trim_pack(const uint8_t* __restrict__ data, uint64_t size, const uint64_t* __restrict__ start, const uint64_t* __restrict__ qstart,
               const uint64_t* __restrict__ qend, uint64_t n, TrimDev T, uint64_t* __restrict__ k16, uint64_t* __restrict__ k48,
               uint64_t* __restrict__ odd, unsigned long long* counters) {
    extern __shared__ uint32_t s_col[];                        // (T.max_alen + 1) * blockDim.x words
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = start[i], qs = qstart[i], qe = qend[i];
    if (s == FQ_NONE) s = size;
    if (qs == FQ_NONE) qs = size;
    if (qe == FQ_NONE) qe = size;
    uint64_t ks[4], kn[2]; int len = 0, oa = 0, ob = 0;
    int rc = trim_record(data, size, s, qs, qe, T, s_col + threadIdx.x, (int)blockDim.x, ks, kn, len, oa, ob);
    if (rc > 2) {
        if (atomicCAS(&counters[2], 0ull, (unsigned long long)rc) == 0ull) counters[3] = i;
        return;
    }
    if (rc == 0) return;
    atomicAdd(&counters[4], 1ull);
    if (rc == 2) {
        odd[atomicAdd(&counters[7], 1ull)] = fq_odd_entry(i, oa, ob - oa);
        atomicAdd(&counters[8], (unsigned long long)(ob - oa));
        return;
    }
    if (!(kn[0] | kn[1]) && len <= 56) {
        unsigned long long p = atomicAdd(&counters[0], 1ull);
        k16[2 * p] = ks[0];
        k16[2 * p + 1] = ks[1] | ((uint64_t)len << 56);
    } else {
        unsigned long long p = atomicAdd(&counters[1], 1ull);
        k48[6 * p + 0] = ks[0]; k48[6 * p + 1] = ks[1]; k48[6 * p + 2] = ks[2];
        k48[6 * p + 3] = ks[3] | ((uint64_t)len << 56);
        k48[6 * p + 4] = kn[0]; k48[6 * p + 5] = kn[1];
    }
}

// phred detection (trim_file.py:123-131): flags[0] = record 0 holds a quality character > 'J', flags[1] = any of
// the first min(n, 1000) records does
__global__ void __launch_bounds__(256)
k_fq_phred(const uint8_t* __restrict__ data, uint64_t size, const uint64_t* __restrict__ qstart, const uint64_t* __restrict__ qend,
           uint64_t n, unsigned long long* flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i >= 1000) return;
    uint64_t qs = qstart[i], qe = qend[i];
    if (qs == FQ_NONE) qs = size;
    if (qe == FQ_NONE) qe = size;
    bool hi = false;
    for (uint64_t p = qs; p < qe; ++p) hi |= data[p] > 74;
    if (hi) { atomicOr(&flags[1], 1ull); if (i == 0) atomicOr(&flags[0], 1ull); }
}
#endif

}  // namespace mg
