// fastq.h -- FASTQ records -> 2-bit read keys, on the device (gfx950).
//
// Replaces the read loop of quantReads (reference miRge.pl:299-307): records are FOUR LINES each, the
// key is line 2 with its "\n" removed (`chomp` strips nothing else, so a "\r" stays in the key and is
// rejected here as a non-nucleotide character); a record cut short by the end of the file still
// counts, with an empty sequence if line 2 is missing.  Header and quality lines are never looked at,
// so '@' or '+' at the start of a quality line is harmless.
//
// The whole file sits in HBM as raw bytes.  Three passes, all streaming:
//   k_fq_count  newlines per 16 KB segment            (reads the file once)
//   scan        exclusive scan of the segment counts  (tiny)
//   k_fq_index  newline ordinal o of every "\n": o % 4 == 0 ends a header line -> start[o / 4] = next byte
//   k_fq_pack   one lane per record: bytes from start[] up to the next "\n" -> 16-B / 48-B key
// No pass relies on bytes at or beyond `size`: loads that begin below `size` stay inside the 16-byte
// (count/index) or 8-byte (pack) aligned unit they begin in, and bytes >= size are masked off.
#pragma once
#include "keys.h"

namespace mg {

constexpr int FQ_TB = 64;                 // bytes per lane in the count / index passes
constexpr int FQ_SEG = 256 * FQ_TB;       // bytes per workgroup
constexpr uint64_t FQ_NONE = ~0ull;       // start[] value of a record whose line 2 does not exist

// high bit of every byte of w that equals '\n' (exact: no borrow between bytes)
MG_HD uint32_t nl_mask32(uint32_t w) {
    uint32_t x = w ^ 0x0A0A0A0Au;
    uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}
// the same, with bytes at file offsets >= size masked off; q = file offset of the word's first byte
MG_HD uint32_t nl_mask32_at(uint32_t w, uint64_t q, uint64_t size) {
    uint32_t m = nl_mask32(w);
    if (q + 4 > size) m = q >= size ? 0u : (m & (0xFFFFFFFFu >> (8 * (int)(4 - (size - q)))));
    return m;
}

// the 16 words of lane-chunk [off, off + 64) -- words that begin at or after `size` read as 0
MG_HD void fq_load64(const uint8_t* __restrict__ data, uint64_t size, uint64_t off, uint32_t (&w)[16]) {
    const uint4* v = reinterpret_cast<const uint4*>(data + off);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint4 x = make_uint4(0, 0, 0, 0);
        if (off + 16 * j < size) x = v[j];
        w[4 * j] = x.x; w[4 * j + 1] = x.y; w[4 * j + 2] = x.z; w[4 * j + 3] = x.w;
    }
}
MG_HD uint32_t fq_count64(const uint32_t (&w)[16], uint64_t off, uint64_t size) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) c += (uint32_t)__builtin_popcount(nl_mask32_at(w[j], off + 4 * j, size));
    return c;
}
// walk the newlines of one lane-chunk; o = ordinal (in the file) of its first newline, file_off = file offset of data[0]
// qstart / qend (nullable, both or none): start and end of every record's quality line (line 4), for the trimming path
MG_HD void fq_index64(const uint32_t (&w)[16], uint64_t off, uint64_t size, uint64_t file_off, uint64_t o, uint64_t* __restrict__ start,
                      uint64_t* __restrict__ qstart = nullptr, uint64_t* __restrict__ qend = nullptr) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t m = nl_mask32_at(w[j], off + 4 * j, size);
        while (m) {
            int b = __builtin_ctz(m) >> 3;                    // byte index inside the word
            m &= m - 1;
            const uint64_t at = file_off + off + 4 * j + b;
            if ((o & 3) == 0) start[o >> 2] = at + 1;
            if (qstart) { if ((o & 3) == 2) qstart[o >> 2] = at + 1; else if ((o & 3) == 3) qend[o >> 2] = at; }
            ++o;
        }
    }
}

// One record: bytes data[s ...] up to "\n" or the end of the file -> key words.
// returns 0 ok, 4 = bad character, 5 = longer than 124 nt.  hasN / len as outputs.
// CR: a "\r" ends the line as well (raw FASTQ goes through cutadapt's reader, which strips line ends; the trimmed
// file quantReads opens is only chomp'ed: there a "\r" stays and is rejected as a non-nucleotide character).
template <bool CR = false>
MG_HD int fq_pack_read(const uint8_t* __restrict__ data, uint64_t size, uint64_t s,
                       uint64_t (&sq)[4], uint64_t (&nm)[2], int& len, bool& hasN) {
    sq[0] = sq[1] = sq[2] = sq[3] = 0; nm[0] = nm[1] = 0; len = 0; hasN = false;
    if (s >= size) return 0;                                   // empty or missing sequence line
    const uint64_t* d64 = reinterpret_cast<const uint64_t*>(data);
    uint64_t wi = s >> 3; const int sh = (int)(s & 7) * 8;
    uint64_t cur = d64[wi];                                    // begins below size
    bool bad = false, done = false, over = false;
    // branch-free inside (selects, not divergent blocks of 64-bit shifts: see the note at make_kent in kernels.h)
    for (int it = 0; it < 16 && !done; ++it) {
        uint64_t nxt = ((wi + 1) << 3) < size ? d64[wi + 1] : 0ull;
        uint64_t v = sh ? (cur >> sh) | (nxt << (64 - sh)) : cur;
        cur = nxt; ++wi;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint8_t ch = (uint8_t)(v >> (8 * b));
            const int pos = it * 8 + b;
            done = done || (s + (uint64_t)pos >= size) || ch == '\n' || (CR && ch == '\r');
            const bool take = !done && pos < 124;
            over |= !done && pos >= 124;
            const int code = base_code(ch);
            bad |= take && code < 0;
            const bool isN = take && code == 4;
            hasN |= isN;
            const uint64_t cc = (take && code >= 0 && code < 4) ? (uint64_t)code : 0ull;
            const int wsel = pos >> 5, bsh = 2 * (pos & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) sq[q] |= (q == wsel) ? cc << bsh : 0ull;
            const uint64_t nbit = isN ? 1ull << (pos & 63) : 0ull;
            nm[0] |= pos < 64 ? nbit : 0ull;
            nm[1] |= pos < 64 ? 0ull : nbit;
            len = take ? pos + 1 : len;
        }
    }
    if (over) return 5;                                        // byte 125 of the line is not its terminator
    return bad ? 4 : 0;
}

// A sequence line that does not fit a key (longer than 124 nt, or holding characters other than A C G T N): its
// length and whether the library takes it at all.  0 = yes (a "verbatim" read: collapsed by exact string on the host,
// matched as N where it is not a/c/g/t), 4 = holds a control character, a blank or a byte >= 0x7F (a "\r" among them on
// the trimmed path, where chomp leaves it in the key, miRge.pl:304), 5 = longer than FQ_MAX_LINE.
constexpr int FQ_MAX_LINE = 1000;         // = MIRGE_MAX_READ_LEN (include/mirge_hip.h)
template <bool CR>
MG_HD int fq_line_scan(const uint8_t* __restrict__ data, uint64_t size, uint64_t s, int& len) {
    len = 0;
    for (uint64_t p = s; p < size; ++p) {
        const uint8_t ch = data[p];
        if (ch == '\n' || (CR && ch == '\r')) break;
        if (ch <= 0x20 || ch >= 0x7F) return 4;
        if (++len > FQ_MAX_LINE) return 5;
    }
    return 0;
}
// verbatim-read list entry: record index | first byte of the read inside the sequence line << 32 | its length << 48
MG_HD uint64_t fq_odd_entry(uint64_t i, int a, int len) { return i | ((uint64_t)a << 32) | ((uint64_t)len << 48); }

#ifdef __HIPCC__
__global__ void __launch_bounds__(256)
k_fq_count(const uint8_t* __restrict__ data, uint64_t size, uint32_t* __restrict__ seg_nl) {
    __shared__ uint32_t red[4];
    uint64_t off = (uint64_t)blockIdx.x * FQ_SEG + (uint64_t)threadIdx.x * FQ_TB;
    uint32_t c = 0;
    if (off < size) { uint32_t w[16]; fq_load64(data, size, off, w); c = fq_count64(w, off, size); }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) seg_nl[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
k_fq_index(const uint8_t* __restrict__ data, uint64_t size, uint64_t file_off, uint64_t o0,
           const uint32_t* __restrict__ seg_base, uint64_t* __restrict__ start, uint64_t* __restrict__ qstart, uint64_t* __restrict__ qend) {
    __shared__ uint32_t buf[256];
    uint64_t off = (uint64_t)blockIdx.x * FQ_SEG + (uint64_t)threadIdx.x * FQ_TB;
    uint32_t w[16];
    uint32_t c = 0;
    if (off < size) { fq_load64(data, size, off, w); c = fq_count64(w, off, size); }
    buf[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = (int)threadIdx.x >= o ? buf[threadIdx.x - o] : 0u;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    // data/size describe one window of the file that begins at byte file_off and after o0 newlines
    if (c) fq_index64(w, off, size, file_off, o0 + seg_base[blockIdx.x] + buf[threadIdx.x] - c, start, qstart, qend);
}

// counters as in k_pack: [0] n16, [1] n48, [2] error code, [3] first offending record; [7] verbatim reads, [8] their bytes
__global__ void __launch_bounds__(256)
k_fq_pack(const uint8_t* __restrict__ data, uint64_t size, const uint64_t* __restrict__ start, uint64_t n,
          uint64_t* __restrict__ k16, uint64_t* __restrict__ k48, uint64_t* __restrict__ odd, unsigned long long* counters) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = start[i];
    uint64_t sq[4], nm[2]; int len; bool hasN;
    int rc = fq_pack_read(data, size, s == FQ_NONE ? size : s, sq, nm, len, hasN);
    if (rc) {                                                  // no key holds it: a verbatim read, or an input error
        rc = fq_line_scan<false>(data, size, s, len);
        if (rc) { if (atomicCAS(&counters[2], 0ull, (unsigned long long)rc) == 0ull) counters[3] = i; return; }
        odd[atomicAdd(&counters[7], 1ull)] = fq_odd_entry(i, 0, len);
        atomicAdd(&counters[8], (unsigned long long)len);
        return;
    }
    if (!hasN && len <= 56) {
        unsigned long long p = atomicAdd(&counters[0], 1ull);
        k16[2 * p] = sq[0];
        k16[2 * p + 1] = sq[1] | ((uint64_t)len << 56);
    } else {
        unsigned long long p = atomicAdd(&counters[1], 1ull);
        k48[6 * p + 0] = sq[0]; k48[6 * p + 1] = sq[1]; k48[6 * p + 2] = sq[2];
        k48[6 * p + 3] = sq[3] | ((uint64_t)len << 56);
        k48[6 * p + 4] = nm[0]; k48[6 * p + 5] = nm[1];
    }
}

// verbatim reads -> one byte pool for the host: tab[j] = offset in the pool << 16 | length
__global__ void __launch_bounds__(256)
k_fq_odd_gather(const uint8_t* __restrict__ data, const uint64_t* __restrict__ start, const uint64_t* __restrict__ odd, uint64_t m,
                uint8_t* __restrict__ pool, uint64_t* __restrict__ tab, unsigned long long* cursor) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint64_t e = odd[j];
    const uint64_t s = start[e & 0xFFFFFFFFull] + ((e >> 32) & 0xFFFF);
    const uint32_t len = (uint32_t)(e >> 48);
    const uint64_t off = atomicAdd(cursor, (unsigned long long)len);
    tab[j] = (off << 16) | len;
    for (uint32_t q = 0; q < len; ++q) pool[off + q] = data[s + q];
}
#endif

}  // namespace mg
