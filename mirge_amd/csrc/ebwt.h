// ebwt.h -- the sequences and names of a library read back out of a bowtie 1 index, so that a seqLibs directory
// that ships only <lib>.*.ebwt (miRge.pl:92-97, 189-192; docs/index.html:36) works without a FASTA beside it
// (SURVEY.md 8f #2).  Host code only; nothing of the FM index itself is used -- the matcher has its own seed index.
//
// UNVALIDATED LAYOUT.  bowtie 1.1.1 is not in the reference tree, no index file exists in the build container and
// none can be made (no bowtie-build), so what follows is the small-index (32-bit) layout as recollected from
// bowtie 1's ebwt.h / reference.cpp, checked only against files written by tests/ebwt_writer.py to the same
// description.  Every size the layout implies is cross-checked, and any disagreement is an error that asks for
// the FASTA instead of a guess:
//   <base>.3.ebwt : int32 1 (byte order mark) | uint32 nrecs | nrecs x { uint32 off, uint32 len, uint8 first }
//                   one record per stretch of unambiguous bases: `off` ambiguous characters precede it, `first`
//                   marks the first stretch of a new reference (trailing ambiguous characters are not recorded)
//   <base>.4.ebwt : the unambiguous bases, 2 bits each (A C G T = 0 1 2 3), four per byte, first base in the low bits
//   <base>.1.ebwt : int32 1 | uint32 len | int32 lineRate | int32 linesPerSide | int32 offRate | int32 ftabChars |
//                   int32 flags (negative: -flags holds COLOR = 2, ENTIRE_REVERSE = 4) | uint32 nPat | nPat x uint32 |
//                   uint32 nFrag | 3 nFrag x uint32 | ebwt[ebwtTotSz] | uint32 zOff | 5 x uint32 fchr |
//                   ftab[(4^ftabChars + 1) x uint32] | eftab[2 ftabChars x uint32] | reference names, '\n'-separated, '\0'
//                   with sideSz = linesPerSide << lineRate, sideBwtSz = sideSz - 8, bwtSz = len / 4 + 1,
//                   ebwtTotSz = ceil(bwtSz / (2 sideBwtSz)) * 2 sideSz; `len` = number of unambiguous bases.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace mg {

struct EbwtBytes {
    std::vector<uint8_t> d; size_t at = 0; bool swap = false, ok = true;
    bool load(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
        d.resize(sz > 0 ? (size_t)sz : 0);
        size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
        fclose(f);
        return got == d.size();
    }
    uint32_t u32() {
        if (at + 4 > d.size()) { ok = false; return 0; }
        uint32_t v = (uint32_t)d[at] | ((uint32_t)d[at + 1] << 8) | ((uint32_t)d[at + 2] << 16) | ((uint32_t)d[at + 3] << 24);
        at += 4;
        return swap ? __builtin_bswap32(v) : v;
    }
    uint8_t u8() { if (at + 1 > d.size()) { ok = false; return 0; } return d[at++]; }
    void skip(uint64_t n) { if (n > d.size() - at) { ok = false; at = d.size(); } else at += (size_t)n; }
    // the int32 1 that opens every file tells the byte order
    bool order_mark() {
        uint32_t one = u32();
        if (one == 1) return ok;
        if (__builtin_bswap32(one) == 1) { swap = true; return ok; }
        return false;
    }
};

// 0 = ok; otherwise err says what did not fit.  seqs = the references back to back ('N' for ambiguous stretches),
// offs = n + 1 offsets into it, names = the full header lines (what `bowtie-inspect -n` prints, miRge.pl:457).
inline int ebwt_extract(const std::string& base, std::vector<std::string>& names, std::string& seqs, std::vector<uint64_t>& offs,
                        std::string& err) {
    names.clear(); seqs.clear(); offs.clear();
    EbwtBytes f3, f4, f1;
    if (!f3.load(base + ".3.ebwt") || !f4.load(base + ".4.ebwt") || !f1.load(base + ".1.ebwt")) {
        err = "cannot read " + base + ".{1,3,4}.ebwt"; return 1;
    }
    const std::string hint = " (layout unvalidated, see ebwt.h: put " + base + ".fa beside the index instead)";
    // ---- .3.ebwt: the stretch records
    if (!f3.order_mark()) { err = base + ".3.ebwt: no byte order mark" + hint; return 2; }
    const uint64_t nrecs = f3.u32();
    if (!f3.ok || f3.d.size() != 8 + nrecs * 9) { err = base + ".3.ebwt: size does not match its record count" + hint; return 2; }
    struct Rec { uint32_t off, len; bool first; };
    std::vector<Rec> recs((size_t)nrecs);
    uint64_t bases = 0, nfirst = 0;
    for (auto& r : recs) { r.off = f3.u32(); r.len = f3.u32(); r.first = f3.u8() != 0; bases += r.len; nfirst += r.first; }
    if (nrecs && !recs[0].first) { err = base + ".3.ebwt: the first record does not start a reference" + hint; return 2; }
    // ---- .4.ebwt: the bases
    if (f4.d.size() != (bases + 3) / 4) { err = base + ".4.ebwt: size does not match the stretch lengths" + hint; return 2; }
    // ---- .1.ebwt: header, then skip to the names
    if (!f1.order_mark()) { err = base + ".1.ebwt: no byte order mark" + hint; return 2; }
    const uint64_t len = f1.u32();
    const int32_t lineRate = (int32_t)f1.u32(), linesPerSide = (int32_t)f1.u32(), offRate = (int32_t)f1.u32(), ftabChars = (int32_t)f1.u32();
    int32_t flags = (int32_t)f1.u32();
    (void)offRate;
    if (!f1.ok || len != bases || lineRate < 3 || lineRate > 16 || linesPerSide < 1 || linesPerSide > 64 || ftabChars < 1 || ftabChars > 14) {
        err = base + ".1.ebwt: header does not agree with the stretch records" + hint; return 2;
    }
    if (flags < 0 && ((-flags) & 2)) { err = base + ": colorspace index, not supported"; return 3; }
    const uint64_t nPat = f1.u32(); f1.skip(nPat * 4);
    const uint64_t nFrag = f1.u32(); f1.skip(nFrag * 12);
    const uint64_t sideSz = (uint64_t)linesPerSide << lineRate, sideBwtSz = sideSz - 8, bwtSz = len / 4 + 1;
    const uint64_t ebwtTotSz = (bwtSz + 2 * sideBwtSz - 1) / (2 * sideBwtSz) * 2 * sideSz;
    f1.skip(ebwtTotSz + 4 + 20 + ((1ull << (2 * ftabChars)) + 1) * 4 + (uint64_t)ftabChars * 2 * 4);
    if (!f1.ok || f1.at >= f1.d.size() || f1.d.back() != 0) { err = base + ".1.ebwt: sections run past the end of the file" + hint; return 2; }
    {
        std::string cur;
        for (size_t i = f1.at; i + 1 < f1.d.size(); ++i) {
            const char ch = (char)f1.d[i];
            if (ch == '\0') { err = base + ".1.ebwt: stray terminator inside the reference names" + hint; return 2; }
            if (ch == '\n') { names.push_back(cur); cur.clear(); } else cur.push_back(ch);
        }
        if (!cur.empty()) names.push_back(cur);
    }
    if (names.size() != nfirst) { err = base + ".1.ebwt: " + std::to_string(names.size()) + " names for " + std::to_string(nfirst) + " references" + hint; return 2; }
    // ---- rebuild
    seqs.reserve((size_t)bases + 16);
    uint64_t b = 0;
    for (const auto& r : recs) {
        if (r.first) offs.push_back(seqs.size());
        seqs.append((size_t)r.off, 'N');
        for (uint32_t i = 0; i < r.len; ++i, ++b) seqs.push_back("ACGT"[(f4.d[(size_t)(b >> 2)] >> (2 * (b & 3))) & 3]);
    }
    offs.push_back(seqs.size());
    return 0;
}

}  // namespace mg
