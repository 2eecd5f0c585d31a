// hostsim.hip -- HOST build of the matcher's __host__ __device__ arithmetic (keys.h, and
// pack_chunk / verify / match_one of kernels.h) so that `pytest -m "not gpu"` can check the bit
// tricks against the oracle without a GPU.  Test harness only: nothing in the product links it,
// and the product library has no such entry point.
#include "../../mirge_amd/csrc/kernels.h"
#include "../../mirge_amd/csrc/fastq.h"
#include "../../mirge_amd/csrc/trim.h"
#include "../../mirge_amd/csrc/ebwt.h"
#include <algorithm>
#include <vector>
#include <cstring>
using namespace mg;

struct SimLib {
    std::vector<uint64_t> text, bad;
    std::vector<uint32_t> koff, ref_start, kbits, lut; int lut_shift = 5; std::vector<KEnt> kent; std::vector<uint64_t> kpre;
    uint32_t T; int kb;
    std::vector<PEnt> slots, pent; std::vector<uint32_t> sbits; TabDesc ctab{0, 0, 0}; std::vector<TabDesc> sch;
    std::vector<CSlot> cslots; std::vector<uint32_t> cpos;
    LibDev dev() const { LibDev L; L.text = text.data(); L.bad = bad.data(); L.koff = koff.data(); L.kent = kent.data(); L.kpre = kpre.data(); L.kbits = kbits.data();
                         L.ref_start = ref_start.data(); L.ref_lut = lut.data(); L.lut_shift = lut_shift; L.n_refs = (uint32_t)ref_start.size() - 1; L.T = T; L.kb = kb;
                         L.slots = slots.empty() ? nullptr : slots.data(); L.sbits = sbits.data(); L.pent = pent.data(); L.ppre = nullptr; L.ctab = ctab;
                         L.nsch = (int)sch.size();
                         for (int j = 0; j < MAX_SCHEMES; ++j) L.sch[j] = j < L.nsch ? sch[j] : TabDesc{0, 0, 0};
                         L.cslots = cslots.empty() ? nullptr : cslots.data(); L.cpos = cpos.data(); L.lds_tw = T / 32 + 3; L.lds_bw = T / 64 + 2; L.lds_sw = 0; L.sep = nullptr; L.rank = nullptr;
                         return L; }
};

extern "C" {

// same layout rules as load_library() in mirge_hip.hip; index built with text_kmer()
// flags bit 0: the both-strand slot tables as well (LibDev::kslot, and the block schemes (4,1), (5,2), (4,2), (3,2)), bit 1: no occupancy bits in front of them,
// bit 2: the tables in their compact form (LibDev::cslots: the LDS path, lds_one)
void* sim_lib_new(int32_t n, const char* seqs, const uint64_t* offs, int32_t kb, int32_t flags) {
    SimLib* S = new SimLib();
    uint64_t T = 0;
    for (int i = 0; i < n; ++i) T += offs[i + 1] - offs[i] + 1;
    S->text.assign(T / 32 + 10, 0); S->bad.assign(T / 64 + 10, 0); S->ref_start.assign(n + 1, 0);
    uint64_t p = 0;
    for (int i = 0; i < n; ++i) {
        S->ref_start[i] = (uint32_t)p;
        for (uint64_t j = offs[i]; j < offs[i + 1]; ++j, ++p) {
            uint8_t ch = (uint8_t)seqs[j]; if (ch >= 'a' && ch <= 'z') ch -= 32;
            int code = base_code(ch);
            if (code >= 0 && code < 4) S->text[p >> 5] |= (uint64_t)code << (2 * (p & 31));
            else S->bad[p >> 6] |= 1ull << (p & 63);
        }
        S->bad[p >> 6] |= 1ull << (p & 63); ++p;
    }
    S->ref_start[n] = (uint32_t)p; S->T = (uint32_t)T; S->kb = kb;
    S->lut_shift = ref_lut_shift(T);
    build_ref_lut(S->ref_start.data(), (uint32_t)n, T, S->lut_shift, S->lut);
    uint64_t nb = 1ull << (2 * kb);
    S->koff.assign(nb + 1, 0); S->kent.assign(T ? T : 1, KEnt{0, 0, 0, 0});
    for (uint32_t q = 0; q < T; ++q) S->koff[text_kmer(S->text.data(), q, kb) + 1]++;
    S->kbits.assign(nb / 32 + 1, 0);
    for (uint64_t b = 0; b < nb; ++b) if (S->koff[b + 1]) S->kbits[b >> 5] |= 1u << (b & 31);
    for (uint64_t b = 0; b < nb; ++b) S->koff[b + 1] += S->koff[b];
    std::vector<uint32_t> cur(S->koff.begin(), S->koff.end() - 1);
    for (uint32_t q = 0; q < T; ++q) S->kent[cur[text_kmer(S->text.data(), q, kb)]++] = make_kent(S->text.data(), q);
    // both-strand slot tables, laid out as build_contig_both() / ensure_scheme() in mirge_hip.hip do
    if ((flags & 1) && T > 0) {
        const bool ubits = !(flags & 2), compact = flags & 4;
        const LibDev LD = S->dev();                                      // text, N bitmap and reference offsets are in place
        auto build_c = [&](int k, uint32_t B, uint32_t mask) {                // as build_both(..., compact) in mirge_hip.hip
            const uint64_t nbk = 1ull << (2 * k), base = S->cpos.size(), s0 = S->cslots.size(), w0 = S->sbits.size();
            S->cpos.resize(base + 2 * (size_t)T); S->cslots.resize(s0 + nbk); S->sbits.resize(w0 + nbk / 32, 0);
            std::vector<uint32_t> off(nbk + 1, 0);
            for (uint32_t q = 0; q < T; ++q) { uint32_t bf, br; both_buckets(S->text.data(), q, k, B, mask, bf, br); off[bf + 1]++; off[br + 1]++; }
            for (uint64_t b = 0; b < nbk; ++b) { if (off[b + 1]) S->sbits[w0 + (b >> 5)] |= 1u << (b & 31); off[b + 1] += off[b]; }
            for (auto& o : off) o += (uint32_t)base;
            std::vector<uint32_t> cur2(off.begin(), off.end() - 1);
            for (uint32_t q = 0; q < T; ++q) {
                uint32_t bf, br; both_buckets(S->text.data(), q, k, B, mask, bf, br);
                S->cpos[cur2[bf]++] = q; S->cpos[cur2[br]++] = q | POS_REV;
            }
            for (uint64_t b = 0; b < nbk; ++b) {                        // as k_cslot_fill
                CSlot e{off[b], off[b + 1]};
                if (e.b == e.a + 1) { e.a = S->cpos[off[b]]; e.b = CS_SINGLE; }
                S->cslots[s0 + b] = e;
            }
        };
        auto build_p = [&](int k, uint32_t B, uint32_t mask) {
            const uint64_t nbk = 1ull << (2 * k), base = S->pent.size(), s0 = S->slots.size(), w0 = S->sbits.size();
            const int K = B ? blocks_W(B) : k;
            S->pent.resize(base + 2 * (size_t)T);
            S->slots.resize(s0 + nbk); S->sbits.resize(w0 + nbk / 32, 0);
            std::vector<uint32_t> off(nbk + 1, 0);
            for (uint32_t q = 0; q < T; ++q) { uint32_t bf, br; both_buckets(S->text.data(), q, k, B, mask, bf, br); off[bf + 1]++; off[br + 1]++; }
            for (uint64_t b = 0; b < nbk; ++b) { if (off[b + 1]) S->sbits[w0 + (b >> 5)] |= 1u << (b & 31); off[b + 1] += off[b]; }
            for (auto& o : off) o += (uint32_t)base;
            std::vector<uint32_t> cur2(off.begin(), off.end() - 1);
            for (uint32_t q = 0; q < T; ++q) {
                uint32_t bf, br; both_buckets(S->text.data(), q, k, B, mask, bf, br);
                S->pent[cur2[bf]++] = make_pent(LD, q, K, false); S->pent[cur2[br]++] = make_pent(LD, q, K, true);
            }
            for (uint64_t b = 0; b < nbk; ++b) {                        // as k_slot_fill
                const uint32_t lo = off[b], hi = off[b + 1];
                PEnt e{}; 
                if (hi == lo) e.pos = SLOT_EMPTY;
                else if (hi == lo + 1) e = S->pent[lo];
                else { e.pos = SLOT_LIST; e.info = lo; e.c[0] = hi; }
                S->slots[s0 + b] = e;
            }
        };
        auto build = [&](int k, uint32_t B, uint32_t mask) { if (compact) build_c(k, B, mask); else build_p(k, B, mask); };
        S->ctab = TabDesc{0, 0, tab_info(kb, 0, 0, ubits)};
        build(kb, 0, 0u);
        const uint32_t BM[7][2] = {{0x4444u, 1}, {0x5555u, 2}, {0x4444u, 2}, {0x3444u, 2}, {0x3344u, 2}, {0x3334u, 2}, {0x3333u, 2}};
        for (int j = 0; j < 7; ++j) {
            const uint32_t B = BM[j][0]; const int m = (int)BM[j][1], kbs = std::min(scheme_max_sel(B, m), kb);
            S->sch.push_back(TabDesc{(uint32_t)(compact ? S->cslots.size() : S->slots.size()), (uint32_t)S->sbits.size(), tab_info(kbs, B, m, ubits)});
            for (int i = 0; i < scheme_nshapes(m); ++i) build(kbs, B, scheme_shape(m, i));
        }
        S->sbits.push_back(0);
    }
    S->kpre.assign(S->kent.size(), 0);
    for (size_t i = 0; i < S->kent.size(); ++i) S->kpre[i] = make_kpre(S->kent[i]);
    return S;
}
void sim_lib_free(void* s) { delete (SimLib*)s; }

// returns key words: 2 (16-byte key), 6 (48-byte key), or -4 / -5 like the pack kernel's error codes
int32_t sim_pack(const char* seq, int32_t len, uint64_t* key) {
    if (len > 124) return -5;
    uint8_t row[128]; memset(row, 0, sizeof row); memcpy(row, seq, (size_t)len);
    uint64_t s[4] = {0, 0, 0, 0}, nm[2] = {0, 0}; bool hasN = false, bad = false;
    for (int c = 0; c < 8; ++c) if (c * 16 < len) {
        uint32_t wv[4]; memcpy(wv, row + 16 * c, 16);
        pack_chunk(wv, c, len, s, nm, hasN, bad);
    }
    if (bad) return -4;
    if (!hasN && len <= 56) { key[0] = s[0]; key[1] = s[1] | ((uint64_t)len << 56); return 2; }
    key[0] = s[0]; key[1] = s[1]; key[2] = s[2]; key[3] = s[3] | ((uint64_t)len << 56); key[4] = nm[0]; key[5] = nm[1];
    return 6;
}

// 1 = aligned (ref, amb filled), 0 = not
int32_t sim_match(void* lib, const uint64_t* key, int32_t kw, const int32_t* round8, int32_t* ref, int32_t* amb) {
    SimLib* S = (SimLib*)lib;
    RoundDev P{round8[0], round8[1], round8[2], round8[3], round8[4], round8[5], round8[6], round8[7]};
    int32_t r = -1; uint32_t a = 0; bool ok;
    if (kw == 2) ok = match_one<2>(key, S->dev(), P, r, a); else ok = match_one<6>(key, S->dev(), P, r, a);
    *ref = ok ? r : -1; *amb = ok ? (int32_t)a : 0;
    return ok ? 1 : 0;
}

// a read no key holds, through match_long (kernels.h): 1 = aligned (ref, amb filled), 0 = not
int32_t sim_match_long(void* lib, const char* seq, int32_t len, const int32_t* round8, int32_t* ref, int32_t* amb) {
    SimLib* S = (SimLib*)lib;
    RoundDev P{round8[0], round8[1], round8[2], round8[3], round8[4], round8[5], round8[6], round8[7]};
    std::vector<uint64_t> rec((size_t)verb_words(len));
    verb_pack(seq, len, rec.data());
    const LibDev L = S->dev();
    uint64_t score, kmax, kmin;
    match_long(rec.data(), len, L, P, score, kmax, kmin);
    *ref = -1; *amb = 0;
    if (!score) return 0;
    *ref = (int32_t)ref_of(L, (uint32_t)(score >> 1));
    *amb = ref_of(L, (uint32_t)kmax) != ref_of(L, 0xFFFFFFFFu - (uint32_t)kmin);
    return 1;
}

// FASTQ text -> keys, lane by lane as k_fq_count / k_fq_index / k_fq_pack do it.  data must be 16-byte aligned and
// readable up to the next multiple of 16.  keys: 6 words per record; kw[i] = 2 or 6, or 0 for a verbatim read (then
// keys[6i] = offset of its first byte, keys[6i+1] = its length).  Returns the record count, or
// -4 / -5 (the pack pass's error codes).  win = bytes per window (the product uses 2^31; tests pass small values).
int64_t sim_fastq(const uint8_t* data, uint64_t size, uint64_t win, uint64_t* keys, int32_t* kw, uint64_t cap) {
    uint64_t newlines = 0;
    std::vector<std::vector<uint32_t>> lane_cnt;
    for (uint64_t w0 = 0; w0 < size; w0 += win) {
        uint64_t wsz = std::min(win, size - w0);
        uint64_t lanes = (wsz + FQ_TB - 1) / FQ_TB;
        lane_cnt.emplace_back(lanes, 0u);
        for (uint64_t l = 0; l < lanes; ++l) {
            uint32_t w[16]; fq_load64(data + w0, wsz, l * FQ_TB, w);
            lane_cnt.back()[l] = fq_count64(w, l * FQ_TB, wsz);
            newlines += lane_cnt.back()[l];
        }
    }
    uint64_t lines = newlines + ((size > 0 && data[size - 1] != '\n') ? 1 : 0);
    uint64_t n = (lines + 3) / 4;
    if (n > cap) return -1;
    std::vector<uint64_t> start(n ? n : 1, FQ_NONE);
    uint64_t o = 0; size_t wi = 0;
    for (uint64_t w0 = 0; w0 < size; w0 += win, ++wi) {
        uint64_t wsz = std::min(win, size - w0);
        for (uint64_t l = 0; l < lane_cnt[wi].size(); ++l) {
            uint32_t w[16]; fq_load64(data + w0, wsz, l * FQ_TB, w);
            if (lane_cnt[wi][l]) fq_index64(w, l * FQ_TB, wsz, w0, o, start.data());
            o += lane_cnt[wi][l];
        }
    }
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t sq[4], nm[2]; int len; bool hasN;
        int rc = fq_pack_read(data, size, start[i] == FQ_NONE ? size : start[i], sq, nm, len, hasN);
        uint64_t* k = keys + 6 * i;
        if (rc) {
            rc = fq_line_scan<false>(data, size, start[i], len);
            if (rc) return -rc;
            k[0] = start[i]; k[1] = (uint64_t)len; kw[i] = 0;
            continue;
        }
        if (!hasN && len <= 56) { k[0] = sq[0]; k[1] = sq[1] | ((uint64_t)len << 56); kw[i] = 2; }
        else { k[0] = sq[0]; k[1] = sq[1]; k[2] = sq[2]; k[3] = sq[3] | ((uint64_t)len << 56); k[4] = nm[0]; k[5] = nm[1]; kw[i] = 6; }
    }
    return (int64_t)n;
}

// Raw FASTQ text through the device code of trim.h, record by record (index pass as sim_fastq, then trim_record).
// keys: 6 words per KEPT record, kw = 2 / 6, or 0 for a verbatim read (k[0] = offset of its first byte, k[1] = length); out[0] = records, out[1] = kept, out[2] = phred base used, out[3] = phred
// reported.  Returns kept, or -4 / -5 / -6 (the kernel's error codes), -1 bad spec / cap.
int64_t sim_trim_fastq(const uint8_t* data, uint64_t size, const char* adapter, int32_t trim5, int32_t trim3,
                       uint64_t* keys, int32_t* kw, uint64_t cap, int64_t* out) {
    TrimDev T; std::string msg;
    if (trim_dev_from_spec(adapter, trim5, trim3, T, msg)) return -1;
    uint64_t newlines = 0;
    for (uint64_t off = 0; off < size; off += FQ_TB) { uint32_t w[16]; fq_load64(data, size, off, w); newlines += fq_count64(w, off, size); }
    uint64_t lines = newlines + ((size > 0 && data[size - 1] != '\n') ? 1 : 0);
    uint64_t n = (lines + 3) / 4;
    std::vector<uint64_t> st(n ? n : 1, FQ_NONE), qs(n ? n : 1, FQ_NONE), qe(n ? n : 1, FQ_NONE);
    uint64_t o = 0;
    for (uint64_t off = 0; off < size; off += FQ_TB) {
        uint32_t w[16]; fq_load64(data, size, off, w);
        uint32_t c = fq_count64(w, off, size);
        if (c) fq_index64(w, off, size, 0, o, st.data(), qs.data(), qe.data());
        o += c;
    }
    bool hi0 = false, hi = false;
    for (uint64_t i = 0; i < n && i < 1000; ++i) {
        uint64_t a = qs[i] == FQ_NONE ? size : qs[i], b = qe[i] == FQ_NONE ? size : qe[i];
        for (uint64_t p = a; p < b; ++p) if (data[p] > 74) { hi = true; if (i == 0) hi0 = true; }
    }
    T.base = hi0 ? 64 : 33;
    out[0] = (int64_t)n; out[2] = T.base; out[3] = hi ? 64 : 33;
    uint64_t kept = 0;
    uint32_t col[TRIM_MAX_ALEN + 1];
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t ks[4], kn[2]; int len = 0, oa = 0, ob = 0;
        int rc = trim_record(data, size, st[i] == FQ_NONE ? size : st[i], qs[i] == FQ_NONE ? size : qs[i], qe[i] == FQ_NONE ? size : qe[i],
                             T, col, 1, ks, kn, len, oa, ob);
        if (rc > 2) return -rc;
        if (rc == 0) continue;
        if (kept >= cap) return -1;
        uint64_t* k = keys + 6 * kept;
        if (rc == 2) { k[0] = st[i] + (uint64_t)oa; k[1] = (uint64_t)(ob - oa); kw[kept] = 0; ++kept; continue; }
        if (!(kn[0] | kn[1]) && len <= 56) { k[0] = ks[0]; k[1] = ks[1] | ((uint64_t)len << 56); kw[kept] = 2; }
        else { k[0] = ks[0]; k[1] = ks[1]; k[2] = ks[2]; k[3] = ks[3] | ((uint64_t)len << 56); k[4] = kn[0]; k[5] = kn[1]; kw[kept] = 6; }
        ++kept;
    }
    out[1] = (int64_t)kept;
    return (int64_t)kept;
}

// newlines among the first `size` bytes, counted as k_fq_count does
uint64_t sim_count_newlines(const uint8_t* data, uint64_t size) {
    uint64_t c = 0;
    for (uint64_t off = 0; off < size; off += FQ_TB) { uint32_t w[16]; fq_load64(data, size, off, w); c += fq_count64(w, off, size); }
    return c;
}

uint32_t sim_key_hash(const uint64_t* key, int32_t kw) { return kw == 2 ? key_hash<2>(key) : key_hash<6>(key); }


// ebwt.h on the host: names ('\n'-joined), sequences and offsets of <base>.{1,3,4}.ebwt.  Returns the number of
// references, or -(error code) with the message in names_out.  Capacities in bytes / entries.
int64_t sim_ebwt_extract(const char* base, char* names_out, uint64_t names_cap, char* seqs_out, uint64_t seqs_cap, uint64_t* offs_out, uint64_t offs_cap) {
    std::vector<std::string> names; std::string seqs, err; std::vector<uint64_t> offs;
    int rc = ebwt_extract(base, names, seqs, offs, err);
    if (rc) { snprintf(names_out, (size_t)names_cap, "%s", err.c_str()); return -rc; }
    std::string joined;
    for (size_t i = 0; i < names.size(); ++i) { if (i) joined.push_back('\n'); joined += names[i]; }
    if (joined.size() + 1 > names_cap || seqs.size() > seqs_cap || offs.size() > offs_cap) return -100;
    memcpy(names_out, joined.c_str(), joined.size() + 1);
    memcpy(seqs_out, seqs.data(), seqs.size());
    for (size_t i = 0; i < offs.size(); ++i) offs_out[i] = offs[i];
    return (int64_t)names.size();
}
}
