// mirge_hip.hip -- host side of libmirge_hip.so: context, library loading + seed-index build,
// orchestration of the collapse / cascade / count-vector kernels, and the C ABI declared in
// include/mirge_hip.h.  There is no CPU implementation of the path in here: every entry point
// that computes launches the kernels of kernels.h and fails if HIP does.
#include "../../include/mirge_hip.h"
#include "kernels.h"
#include "fastq.h"
#include "shard.h"
#include "trim.h"
#include "ebwt.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

#include <map>
#include <fcntl.h>
#include <zlib.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace mg;

namespace {

thread_local std::string g_open_error;

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep, size_t keep_bytes, hipStream_t st) {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = std::max(bytes, cap + cap / 2);
        ncap = (ncap + 255) & ~size_t(255);
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && keep_bytes) {
            e = hipMemcpyAsync(q, p, std::min(keep_bytes, cap), hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        }
        if (p) (void)hipFree(p);
        p = q; cap = ncap;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- workspace of the one entry point that allocates nothing (mirge_collapse_keys16_noalloc).  A fixed array in the
// code object: every GPU the library runs on holds its own copy, at its own address, until the process ends.  Host code
// never passes the array's NAME to a kernel or a memset (on the host that name is a stand-in, not a device address): the
// address of the current device's copy is asked for once per device (na_scratch_init, called by mirge_open) and kept.
constexpr uint64_t NA_MAX_KEYS = MIRGE_NOALLOC_MAX_KEYS;
constexpr uint64_t NA_SLOTS = 2 * NA_MAX_KEYS;                  // power of two >= 1.25 * NA_MAX_KEYS: load <= 0.5 at the maximum
struct NaLayout {
    uint32_t lslots[NA_SLOTS];                                  // k_collapse_local's open-addressing table
    uint32_t cnt[NA_MAX_KEYS], cnt0[NA_MAX_KEYS];               // duplicate adds / the claimant's own count per input key
    uint32_t coo_g[NA_MAX_KEYS];                                // k_collapse_emit's unique ids (0, 1, 2, ... here: nothing is known beforehand)
    unsigned long long ctr[32];                                 // k_collapse_emit's packed rank counter
};
static __device__ __align__(256) unsigned char scratch[sizeof(NaLayout)];
constexpr int NA_MAX_DEVICES = 64;
std::mutex g_na_mu;
NaLayout* g_na_ptr[NA_MAX_DEVICES] = {};                        // per device: where that GPU's copy of `scratch` lives
// with `device` current
hipError_t na_scratch_init(int device) {
    if (device < 0 || device >= NA_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_na_mu);
    if (g_na_ptr[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_na_ptr[device] = reinterpret_cast<NaLayout*>(p);
    return hipSuccess;
}

struct Lib {
    bool loaded = false;
    std::vector<std::string> names, short_names;
    std::vector<uint32_t> ref_start;     // n + 1
    uint64_t T = 0;
    int kb = 0;
    DevBuf text, bad, koff, kent, kpre, kbits, dref, dlut; int lut_shift = 5;
    bool use_bits = true;                // false: koff is small enough to stay in L2, the occupancy bitmap is skipped
    // both-strand slot tables (LibDev::slots ...): the contiguous one is built with the library, the block-scheme ones
    // the first time a cascade holds a round that can use them (ensure_schemes); all in two buffers
    DevBuf slots, sbits, pent, ppre; uint64_t n_slots = 0, n_bitw = 0, n_pent = 0; bool has_both = false;
    // compact form of the same tables (kernels.h: LibDev::cslots) for a library whose text fits the LDS: 8-byte slots,
    // 4-byte entries; n_slots / n_bitw / n_pent, ctab and sch then count and describe these
    DevBuf cslots, cpos, sep, rank; bool has_compact = false; uint32_t lds_tw = 0, lds_bw = 0, lds_sw = 0;
    std::vector<uint64_t> h_sep; bool has_n = false;         // host: the separators' bitmap (until the compact tables are built), any reference N?
    TabDesc ctab{0, 0, 0}; std::vector<TabDesc> sch;
    LibDev dev() const {
        LibDev L;
        L.text = text.as<uint64_t>(); L.bad = bad.as<uint64_t>(); L.koff = koff.as<uint32_t>();
        L.kent = kent.as<KEnt>(); L.kpre = kpre.as<uint64_t>(); L.kbits = use_bits ? kbits.as<uint32_t>() : nullptr; L.ref_start = dref.as<uint32_t>();
        L.ref_lut = dlut.as<uint32_t>(); L.lut_shift = lut_shift;
        L.n_refs = (uint32_t)names.size(); L.T = (uint32_t)T; L.kb = kb;
        L.slots = has_both ? slots.as<PEnt>() : nullptr; L.sbits = sbits.as<uint32_t>(); L.pent = pent.as<PEnt>(); L.ppre = ppre.as<uint64_t>(); L.ctab = ctab;
        L.nsch = (int)sch.size();
        for (int j = 0; j < MAX_SCHEMES; ++j) L.sch[j] = j < L.nsch ? sch[j] : TabDesc{0, 0, 0};
        L.cslots = has_compact ? cslots.as<CSlot>() : nullptr; L.cpos = cpos.as<uint32_t>();
        L.lds_tw = has_compact ? lds_tw : 0; L.lds_bw = has_compact ? lds_bw : 0; L.lds_sw = has_compact ? lds_sw : 0;
        L.sep = lds_sw ? sep.as<uint64_t>() : nullptr; L.rank = rank.as<uint32_t>();
        return L;
    }
    void release() { cslots.release(); cpos.release(); sep.release(); rank.release(); has_compact = false; text.release(); bad.release(); koff.release(); kent.release(); kpre.release(); kbits.release(); dref.release(); dlut.release(); slots.release(); sbits.release(); pent.release(); ppre.release(); sch.clear(); n_slots = n_bitw = n_pent = 0; has_both = false; loaded = false; }
};

constexpr size_t STAT_WORDS = 2 * STAT_SLOTS * 8;         // uint64 words of counters per round
constexpr int LL_SPLIT = 26;                              // miRge.pl:326: round 1 takes length < 26, round 2 length > 25
constexpr int LEN_BINS = MIRGE_MAX_READ_LEN + 1;          // read-length histogram rows (miRge.pl:315)

struct Segment { int sample; uint64_t start, count; };

// The unique table of one key class (16-byte or 48-byte keys).
struct Class {
    int KW = 2;
    uint64_t U = 0;             // uniques so far (all samples)
    uint64_t coo_n = 0;         // (unique, count) entries so far
    DevBuf ukeys, coo_g, coo_c;
    DevBuf gslots; uint64_t gcap = 0; bool gvalid = false;     // cross-sample table, built lazily
    DevBuf ann_round, ann_ref, ann_amb;
    DevBuf act[2];                              // ping-pong lists of still-unannotated uniques
    unsigned long long* act_n = nullptr;        // their lengths (device; lives in the context's state arena)
    DevBuf fb; unsigned long long* fb_n = nullptr;      // reads the probe kernel hands to the generic kernel, per round (16-byte class)
    // what the generic kernel's launches over `fb` leave unannotated: ping-pong dense lists of their own, read by the next
    // round's launch of that kernel (the probe kernel's survivors go to split lists: kernels.h, ListOut)
    DevBuf fbs[2]; unsigned long long* fbs_n = nullptr;
    struct ListFmt { uint32_t K = 0, cap = 0; } lf[MIRGE_MAX_ROUNDS + 1];      // layout of the list round r reads (K <= 1: dense)
    // The two classes run side by side (the 48-byte class is tiny: its kernels are latency-, not throughput-bound):
    // each has its own stream and its own collapse workspaces.
    hipStream_t st = nullptr;
    DevBuf ws_lslots, ws_cnt, ws_cnt0, ws_ctr;
    unsigned long long* h_ctr = nullptr;        // pinned, 4 words
    uint32_t pend_n = 0; int pend_sample = 0; bool pending = false;     // a collapse launched, not yet finished
    bool pend_part = false; const uint64_t* pend_keys = nullptr; const uint32_t* pend_wt = nullptr;      // ... in partitioned form: its inputs, should it have to start over
    DevBuf pt_k1, pt_w1, pt_k2, pt_w2, pt_cur;                          // partitioned collapse: level-1 / level-2 regions, cursors + overflow word
    // length-class lists (16-byte class): ids of the uniques shorter than LL_SPLIT nt / of the others, appended by
    // k_part_fold.  Complete -- and walked by rounds 1 and 2 of the reference cascade -- as long as every unique since
    // the last reset came through the partitioned form (ll_ok); ll_fresh: their counters were zeroed since that reset.
    DevBuf ll[2], ll_ctr; bool ll_ok = true, ll_fresh = false;
    std::vector<Segment> segs;
    void reset() { U = 0; coo_n = 0; gvalid = false; segs.clear(); ll_ok = true; ll_fresh = false; }
    void release() { ukeys.release(); coo_g.release(); coo_c.release(); gslots.release(); act[0].release(); act[1].release(); fb.release(); fbs[0].release(); fbs[1].release(); ann_round.release(); ann_ref.release(); ann_amb.release();
                     ws_lslots.release(); ws_cnt.release(); ws_cnt0.release(); ws_ctr.release();
                     pt_k1.release(); pt_w1.release(); pt_k2.release(); pt_w2.release(); pt_cur.release(); ll[0].release(); ll[1].release(); ll_ctr.release(); }
};

// Verbatim uniques (kernels.h: match_long): reads no key can hold, collapsed by their exact string as miRge.pl:306 does.
// A host-side table.  In small-RNA data they are the odd read, but a run can be made of nothing else (150-nt reads,
// --adapter none), so a batch of them is added by the host's threads: the table is cut into NSH shards by the string's
// hash, phase 1 hashes the batch's strings (threads over slices of the batch), phase 2 gives every shard to one thread,
// which looks its strings up in the shard's open-addressing table (64-bit hash, then the bytes) in batch order, phase 3
// numbers what is new.  Ids and entries come out exactly as the serial loop numbers them -- by first occurrence -- because
// "new" is a flag per batch item and the numbers are its prefix sums.
struct Verb {
    static constexpr int NSH = 64;
    static constexpr uint32_t NONE = 0xFFFFFFFFu, PEND = 0x80000000u;
    struct Ent { uint64_t h; uint32_t id, tix; uint64_t epoch; };       // id: index into seq (PEND | k while its batch runs); tix: its line in the batch's touch list
    struct Touch { uint32_t id, first, count, pos; };                      // a unique the running batch met: where first, how often, its entry
    struct Shard {
        std::vector<Ent> tab; size_t used = 0;
        std::unordered_map<uint64_t, uint32_t> entry;                      // unique * S + sample -> position in the entry arrays
        std::vector<std::string> pend; std::vector<uint32_t> pend_slot;    // the running batch's new strings
        std::vector<Touch> touch;
    };
    Shard sh[NSH];
    std::vector<std::string> seq;
    std::vector<uint32_t> coo_u, coo_s, coo_c;
    uint32_t maxlen = 0;
    uint64_t epoch = 0;
    DevBuf pool, off, len, ann_round, ann_ref, ann_amb, d_u, d_s, d_c;
    void reset() {
        for (auto& x : sh) { x.tab.clear(); x.used = 0; x.entry.clear(); x.pend.clear(); x.pend_slot.clear(); x.touch.clear(); }
        seq.clear(); coo_u.clear(); coo_s.clear(); coo_c.clear(); maxlen = 0;
    }
    void release() { pool.release(); off.release(); len.release(); ann_round.release(); ann_ref.release(); ann_amb.release(); d_u.release(); d_s.release(); d_c.release(); }
    static uint64_t hash(const char* s, size_t n) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, s + i, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
        uint64_t w = 0; if (i < n) memcpy(&w, s + i, n - i);
        h = (h ^ w) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 32;
        return h;
    }
    static int shard_of(uint64_t h) { return (int)(h >> 58); }             // (NSH = 64; the table slots take the low bits)
    const std::string& str_of(const Shard& x, uint32_t id) const { return (id & PEND) ? x.pend[id & ~PEND] : seq[id]; }
    void grow(Shard& x) {
        const size_t cap = x.tab.empty() ? 1024 : x.tab.size() * 2;
        std::vector<Ent> nt(cap, Ent{0, NONE, 0, 0});
        for (const Ent& e : x.tab) if (e.id != NONE) { size_t k = (size_t)e.h & (cap - 1); while (nt[k].id != NONE) k = (k + 1) & (cap - 1); nt[k] = e; }
        x.tab.swap(nt);
        for (size_t k = 0; k < x.tab.size(); ++k) if (x.tab[k].id != NONE && (x.tab[k].id & PEND)) x.pend_slot[x.tab[k].id & ~PEND] = (uint32_t)k;
    }
    // the slot of string s in shard x (its id is NONE if the string is new: the caller fills it in)
    size_t find(Shard& x, uint64_t h, const char* s, size_t n) {
        if ((x.used + 1) * 2 > x.tab.size()) grow(x);
        const size_t cap = x.tab.size();
        size_t k = (size_t)h & (cap - 1);
        for (;; k = (k + 1) & (cap - 1)) {
            Ent& e = x.tab[k];
            if (e.id == NONE) return k;
            if (e.h == h) { const std::string& t = str_of(x, e.id); if (t.size() == n && memcmp(t.data(), s, n) == 0) return k; }
        }
    }
    // one string (imports, small batches)
    void add(const char* s, size_t n, int sample, int S, uint32_t count) {
        const uint64_t h = hash(s, n);
        Shard& x = sh[shard_of(h)];
        const size_t k = find(x, h, s, n);
        if (x.tab[k].id == NONE) { x.tab[k] = Ent{h, (uint32_t)seq.size(), 0, 0}; ++x.used; seq.emplace_back(s, n); if (n > maxlen) maxlen = (uint32_t)n; }
        const uint32_t u = x.tab[k].id;
        auto e = x.entry.emplace((uint64_t)u * (uint64_t)S + (uint64_t)sample, (uint32_t)coo_u.size());
        if (e.second) { coo_u.push_back(u); coo_s.push_back((uint32_t)sample); coo_c.push_back(count); }
        else coo_c[e.first->second] += count;
    }
    // a batch of one sample's strings, once each: item(i) -> (pointer, length)
    template <class Item> void add_batch(uint64_t n, Item item, int sample, int S, int nthr) {
        if (n < 8192 || nthr <= 1 || n >= 0x7FFFFFFFull) { for (uint64_t i = 0; i < n; ++i) { auto it = item(i); add(it.first, it.second, sample, S, 1); } return; }
        ++epoch;
        auto run = [&](int nt, auto fn) { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(t); }); for (auto& q : th) q.join(); };
        // phase 1: hashes; every thread lists its slice's items by shard (in batch order)
        std::vector<uint64_t> hs((size_t)n);
        std::vector<std::vector<std::vector<uint32_t>>> lists((size_t)nthr, std::vector<std::vector<uint32_t>>(NSH));
        std::vector<uint32_t> tmax((size_t)nthr, 0);
        run(nthr, [&](int t) {
            uint32_t mx = 0;
            for (uint64_t i = n * t / nthr, e = n * (t + 1) / nthr; i < e; ++i) {
                auto it = item(i);
                const uint64_t h = hash(it.first, it.second);
                hs[(size_t)i] = h; lists[(size_t)t][(size_t)shard_of(h)].push_back((uint32_t)i);
                if (it.second > mx) mx = (uint32_t)it.second;
            }
            tmax[(size_t)t] = mx;
        });
        for (uint32_t m : tmax) if (m > maxlen) maxlen = m;
        // phase 2: a shard at a time per thread; new strings and new (unique, sample) pairs are flagged per batch item
        std::vector<uint8_t> newu((size_t)n, 0), newe((size_t)n, 0);
        std::atomic<int> next{0};
        run(nthr, [&](int) {
            for (;;) {
                const int si = next.fetch_add(1);
                if (si >= NSH) return;
                Shard& x = sh[si];
                x.pend.clear(); x.pend_slot.clear(); x.touch.clear();
                for (int t = 0; t < nthr; ++t) for (uint32_t i : lists[(size_t)t][(size_t)si]) {
                    auto it = item(i);
                    const size_t k = find(x, hs[i], it.first, it.second);
                    Ent& e = x.tab[k];
                    if (e.id == NONE) {
                        e = Ent{hs[i], PEND | (uint32_t)x.pend.size(), 0, 0}; ++x.used;
                        x.pend.emplace_back(it.first, it.second); x.pend_slot.push_back((uint32_t)k);
                        newu[i] = 1;
                    }
                    if (e.epoch != epoch) {
                        e.epoch = epoch; e.tix = (uint32_t)x.touch.size();
                        uint32_t pos = NONE;
                        if (!(e.id & PEND)) { auto f = x.entry.find((uint64_t)e.id * (uint64_t)S + (uint64_t)sample); if (f != x.entry.end()) pos = f->second; }
                        if (pos == NONE) newe[i] = 1;
                        x.touch.push_back(Touch{e.id, i, 0, pos});
                    }
                    ++x.touch[e.tix].count;
                }
            }
        });
        // phase 3: numbers = prefix sums of the flags (the order a serial loop would give), then every shard files its own
        std::vector<uint32_t> ru((size_t)n), re((size_t)n);
        uint32_t nu = 0, ne = 0;
        for (uint64_t i = 0; i < n; ++i) { ru[(size_t)i] = nu; nu += newu[(size_t)i]; re[(size_t)i] = ne; ne += newe[(size_t)i]; }
        const uint32_t u0 = (uint32_t)seq.size(), e0 = (uint32_t)coo_u.size();
        seq.resize((size_t)u0 + nu); coo_u.resize((size_t)e0 + ne); coo_s.resize((size_t)e0 + ne); coo_c.resize((size_t)e0 + ne);
        next = 0;
        run(nthr, [&](int) {
            for (;;) {
                const int si = next.fetch_add(1);
                if (si >= NSH) return;
                Shard& x = sh[si];
                for (Touch& t : x.touch) {
                    uint32_t u = t.id;
                    if (u & PEND) {
                        const uint32_t k = u & ~PEND;
                        u = u0 + ru[t.first];
                        seq[u] = std::move(x.pend[k]);
                        x.tab[x.pend_slot[k]].id = u;
                    }
                    if (t.pos == NONE) {
                        const uint32_t pos = e0 + re[t.first];
                        coo_u[pos] = u; coo_s[pos] = (uint32_t)sample; coo_c[pos] = t.count;
                        x.entry.emplace((uint64_t)u * (uint64_t)S + (uint64_t)sample, pos);
                    } else coo_c[t.pos] += t.count;
                }
                x.pend.clear(); x.pend_slot.clear(); x.touch.clear();
            }
        });
    }
};

struct KTime { const char* name; hipEvent_t a = nullptr, b = nullptr; bool used = false, ok = false; double ms = 0; uint64_t bytes = 0; };

}  // namespace

struct mirge_ctx {
    int device = 0;
    int n_cu = 256;                             // compute units of the device (grid sizes of the striding kernels)
    int S = 1;
    hipStream_t st = nullptr, st2 = nullptr;        // st: everything; st2: the 48-byte key class inside collapse_packed / annotate
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_c[2] = {nullptr, nullptr};           // timing marks of the last collapse call (its own pair: a fused call runs the cascade before they are read)
    hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};   // timing marks of the last collapse / cascade call (made once: no per-call create, no leak on an error return)
    std::string err;
    Lib libs[MIRGE_N_LIBS];
    int n_rounds = 5;
    int coop_thr = 4;                               // candidate lists longer than this are drained by whole waves (MIRGE_COOP_THR)
    int coop_thr3 = 16;                             // the same for rounds cut into >= 3 pieces (-n 2): their many long lists would
                                                    // overflow the LDS queue, so more of them stay with their thread (MIRGE_COOP_THR3)
    mirge_round rounds[MIRGE_MAX_ROUNDS];
    Class cls[2];                                   // [0] 16-byte keys, [1] 48-byte keys
    Verb vb;                                        // the reads neither holds
    DevBuf fq_odd, fq_oddtab, fq_oddpool;           // FASTQ ingest: verbatim-read list, their bytes for the host
    // workspaces
    DevBuf ctr, stage16, stage48, in_seqs, in_lens, scan_tmp, hist, off_tmp;
    DevBuf sh_ctr, sh_w[2];                         // multi-GPU exchange: per-owner / per-sample counters, weights of imported records
    int sh_world = 0; uint64_t sh_n[2][SHARD_MAX_WORLD] = {};
    DevBuf fq_raw, fq_seg, fq_segb, fq_start, fq_qstart, fq_qend;       // FASTQ ingest: raw file bytes, newline counts per segment (+ scanned), record starts
    uint8_t* fq_pin = nullptr;                      // pinned staging ring for file -> HBM streaming (FQ_SLOTS x FQ_SLOT_BYTES)
    double ingest_ms = 0;                           // wall time of the last file -> HBM transfer
    // everything a run starts from zero and hands back lives in ONE arena: [round stats | quant | iscan | categories |
    // length histogram | trimmedUniq | survivor-list lengths x 2] -- one memset before the cascade, one copy to the host after it
    DevBuf state; unsigned long long* h_state = nullptr; size_t h_state_cap = 0;
    uint8_t* h_odd = nullptr; size_t h_odd_cap = 0;     // pinned: the verbatim reads of a FASTQ call on their way to the host table
    unsigned long long *d_stats = nullptr, *d_quant = nullptr, *d_iscan = nullptr, *d_cat = nullptr, *d_len = nullptr, *d_tu = nullptr;
    unsigned long long* h_ctr = nullptr;            // pinned, 16 words
    // results
    bool annotated = false, counts_valid = false, host_valid = false;
    uint64_t stats[MIRGE_MAX_ROUNDS][2] = {};
    uint64_t handed_on[MIRGE_MAX_ROUNDS] = {};      // per round: uniques the probe / LDS kernel handed to the generic kernel (both key classes)
    double round_ms[MIRGE_MAX_ROUNDS] = {};
    std::vector<uint64_t> h_quant, h_iscan, h_cat, h_len, h_tu;
    // host copies for per-unique export
    std::vector<uint64_t> h_keys[2];
    std::vector<uint8_t> h_round[3], h_amb[3];      // [2]: the verbatim uniques
    std::vector<int32_t> h_ref[3];
    std::vector<uint32_t> h_coo_g[2], h_coo_c[2];
    std::vector<uint64_t> csr_off; std::vector<uint32_t> csr_sample, csr_count;
    std::string uniq_seq; std::vector<uint32_t> uniq_counts;
    // timing
    std::vector<KTime> kt; bool timing = true;          // mirge_set_kernel_timing
    double collapse_ms = 0, annotate_ms = 0, counts_ms = 0;
};

namespace {

int fail(mirge_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_open_error = msg;
    return code;
}
#define HIPCHK(c, call)                                                                         \
    do { hipError_t e__ = (call);                                                               \
         if (e__ != hipSuccess) return fail((c), MIRGE_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

KTime& ktime(mirge_ctx* c, const char* name) {
    for (auto& k : c->kt) if (k.name == name || !strcmp(k.name, name)) return k;
    KTime k; k.name = name;
    if (hipEventCreate(&k.a) != hipSuccess) k.a = nullptr;
    if (hipEventCreate(&k.b) != hipSuccess) k.b = nullptr;
    c->kt.push_back(k);
    return c->kt.back();
}
// a timing mark that could not be recorded leaves the entry unused (no time is reported for it) instead of a stale one
void kt_begin(mirge_ctx* c, const char* name, hipStream_t st = nullptr) { if (!c->timing) return; KTime& k = ktime(c, name); k.ok = k.a && k.b && hipEventRecord(k.a, st ? st : c->st) == hipSuccess; }
void kt_end(mirge_ctx* c, const char* name, uint64_t bytes, hipStream_t st = nullptr) {
    if (!c->timing) return;
    KTime& k = ktime(c, name);
    k.ok = k.ok && hipEventRecord(k.b, st ? st : c->st) == hipSuccess;
    k.used = k.ok; k.bytes = bytes; k.ms = -1;
}
// st2 picks up after everything queued on st so far / st waits for everything queued on st2 so far
int fork_st2(mirge_ctx* c) {
    if (hipEventRecord(c->ev_fork, c->st) != hipSuccess || hipStreamWaitEvent(c->st2, c->ev_fork, 0) != hipSuccess) return fail(c, MIRGE_E_HIP, "stream fork failed");
    return MIRGE_OK;
}
int join_st2(mirge_ctx* c) {
    if (hipEventRecord(c->ev_join, c->st2) != hipSuccess || hipStreamWaitEvent(c->st, c->ev_join, 0) != hipSuccess) return fail(c, MIRGE_E_HIP, "stream join failed");
    return MIRGE_OK;
}
void kt_resolve(mirge_ctx* c) {
    for (auto& k : c->kt) if (k.used && k.ms < 0) { float ms = 0; if (hipEventElapsedTime(&ms, k.a, k.b) == hipSuccess) k.ms = ms; else k.ms = 0; }
}
void kt_clear(mirge_ctx* c, const char* prefix) {
    for (auto& k : c->kt) if (!strncmp(k.name, prefix, strlen(prefix))) { k.used = false; k.ms = 0; }
}

uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1024; while (p < x) p <<= 1; return p; }

// ---- exclusive scan of n uint32 (in -> out), using c->scan_tmp
int scan_exclusive(mirge_ctx* c, const uint32_t* in, uint32_t* out, uint64_t n) {
    uint32_t m = (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    HIPCHK(c, c->scan_tmp.ensure((size_t)m * 4 + 16, false, 0, c->st));
    hipLaunchKernelGGL(k_scan_sums, dim3(m), dim3(256), 0, c->st, in, n, c->scan_tmp.as<uint32_t>());
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->st, c->scan_tmp.as<uint32_t>(), m);
    hipLaunchKernelGGL(k_scan_apply, dim3(m), dim3(256), 0, c->st, in, n, c->scan_tmp.as<uint32_t>(), out);
    HIPCHK(c, hipGetLastError());
    return MIRGE_OK;
}

// ---- both-strand slot tables (kernels.h: LibDev::kslot, both_buckets)
__global__ void __launch_bounds__(256) k_add_u32(uint32_t* a, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += v;
}
bool want_slots() { const char* env = getenv("MIRGE_SLOTS"); return !(env && atoi(env) == 0); }      // A/B aid
bool bits_wanted(uint64_t nbk) {                 // an occupancy bitmap in front of a slot table pays while it stays in the L2
    uint64_t lim = 4ull << 20;
    if (const char* env = getenv("MIRGE_BITS2_MAX")) lim = strtoull(env, nullptr, 10);
    return nbk / 8 <= lim;
}
// One table: 2T entries (every position under its own key and under the reverse strand's) appended to `pent`, nbk = 4^k
// slots and nbk / 32 occupancy words appended to the library's slot / occupancy buffers (all grown here).
// B = 0: the contiguous k-mer table; else the table of one shape (block mask) of a block scheme.
int build_both(mirge_ctx* c, Lib& L, int k, uint32_t B, uint32_t mask, bool compact = false) {
    const uint64_t nbk = 1ull << (2 * k), T = L.T, base = L.n_pent;
    if (base + 2 * T >= 0xFFFFFFF0ull || L.n_slots + nbk >= 0xFFFFFFF0ull || L.n_bitw + nbk / 32 >= (1ull << 27))
        return fail(c, MIRGE_E_LIMIT, "seed tables of a library exceed 2^32 entries");
    if (compact) {
        HIPCHK(c, L.cpos.ensure((base + 2 * T) * 4, true, base * 4, c->st));
        HIPCHK(c, L.cslots.ensure((L.n_slots + nbk) * sizeof(CSlot), true, L.n_slots * sizeof(CSlot), c->st));
    } else {
    HIPCHK(c, L.pent.ensure((base + 2 * T) * sizeof(PEnt), true, base * sizeof(PEnt), c->st));
    HIPCHK(c, L.ppre.ensure((base + 2 * T) * 8, true, base * 8, c->st));
    HIPCHK(c, L.slots.ensure((L.n_slots + nbk) * sizeof(PEnt), true, L.n_slots * sizeof(PEnt), c->st));
    }
    HIPCHK(c, L.sbits.ensure((L.n_bitw + nbk / 32 + 1) * 4, true, L.n_bitw * 4, c->st));
    HIPCHK(c, c->hist.ensure((nbk + 1) * 4, false, 0, c->st));
    HIPCHK(c, c->off_tmp.ensure((nbk + 1) * 4, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->hist.p, 0, (nbk + 1) * 4, c->st));
    const uint32_t grid = (uint32_t)((T + 255) / 256);
    hipLaunchKernelGGL(k_both_hist, dim3(grid), dim3(256), 0, c->st, L.text.as<uint64_t>(), (uint32_t)T, k, B, mask, c->hist.as<uint32_t>());
    hipLaunchKernelGGL(k_bucket_bits, dim3((uint32_t)((nbk / 32 + 256) / 256)), dim3(256), 0, c->st, c->hist.as<uint32_t>(), nbk, L.sbits.as<uint32_t>() + L.n_bitw);
    int rc = scan_exclusive(c, c->hist.as<uint32_t>(), c->off_tmp.as<uint32_t>(), nbk + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_add_u32, dim3((uint32_t)((nbk + 1 + 255) / 256)), dim3(256), 0, c->st, c->off_tmp.as<uint32_t>(), nbk + 1, (uint32_t)base);
    HIPCHK(c, hipMemcpyAsync(c->hist.p, c->off_tmp.p, nbk * 4, hipMemcpyDeviceToDevice, c->st));
    if (compact) {
        hipLaunchKernelGGL(k_both_scatter_pos, dim3(grid), dim3(256), 0, c->st, L.text.as<uint64_t>(), (uint32_t)T, k, B, mask, c->hist.as<uint32_t>(), L.cpos.as<uint32_t>());
        hipLaunchKernelGGL(k_cslot_fill, dim3((uint32_t)((nbk + 255) / 256)), dim3(256), 0, c->st, c->off_tmp.as<uint32_t>(), L.cpos.as<uint32_t>(), nbk, L.cslots.as<CSlot>() + L.n_slots);
    } else {
    hipLaunchKernelGGL(k_both_scatter, dim3(grid), dim3(256), 0, c->st, L.dev(), k, B, mask, c->hist.as<uint32_t>(), L.pent.as<PEnt>());
    hipLaunchKernelGGL(k_slot_fill, dim3((uint32_t)((nbk + 255) / 256)), dim3(256), 0, c->st, c->off_tmp.as<uint32_t>(), L.pent.as<PEnt>(), nbk, L.slots.as<PEnt>() + L.n_slots);
    hipLaunchKernelGGL(k_pent_pre, dim3((uint32_t)((2 * T + 255) / 256)), dim3(256), 0, c->st, L.pent.as<PEnt>() + base, 2 * T, L.ppre.as<uint64_t>() + base);
    }
    HIPCHK(c, hipGetLastError());
    L.n_pent = base + 2 * T; L.n_slots += nbk; L.n_bitw += nbk / 32;
    return MIRGE_OK;
}
// the contiguous table, with the library
int build_contig_both(mirge_ctx* c, Lib& L) {
    if (L.T == 0 || L.names.size() > PENT_MAX_REFS) return MIRGE_OK;      // an entry holds its reference in 20 bits
    const uint64_t nbk = 1ull << (2 * L.kb);
    L.ctab = TabDesc{(uint32_t)L.n_slots, (uint32_t)L.n_bitw, tab_info(L.kb, 0, 0, bits_wanted(nbk))};
    int rc = build_both(c, L, L.kb, 0, 0u);
    if (rc) return rc;
    L.has_both = true;
    return MIRGE_OK;
}
// A library whose text + N bitmap a block can stage into LDS next to its waves' queues CAN take the compact tables and
// k_lds_round: MIRGE_LDS = 1.  Off by default -- measured on MI355X (profiles/r03/lds_path.txt) the path loses to the
// probe kernel in time (round 5: 0.57 against 0.40 ms, round 1: 0.15 against 0.13) and moves the same bytes past the L2
// (2.3 GB in round 5 either way: what the L2 misses are the random 128-B lines of slots and lists, whose count does
// not depend on the entry size).  MIRGE_LDS_MAX = bytes of text + bitmap (default 40 KB: two blocks per CU).
uint64_t lds_lib_max() { if (const char* e = getenv("MIRGE_LDS_MAX")) return strtoull(e, nullptr, 10); return 40u << 10; }
bool lds_wanted(const Lib& L) {
    { const char* e = getenv("MIRGE_LDS"); if (!e || atoi(e) == 0) return false; }
    if (L.T == 0) return false;
    const uint64_t tw = L.T / 32 + 3, bw = L.T / 64 + 2;
    // text + bitmap(s) + rank words; entries of every table the library can get: the contiguous one + schemes (4,1), (5,2), (4,2), (3,2) = 1 + 4 + 18
    return (tw + bw * (L.has_n ? 2 : 1)) * 8 + bw * 4 <= lds_lib_max() && 2 * L.T * 23 < LDS_MAX_ENTRIES && L.T < (1u << 24);
}
size_t lds_lib_bytes(const LibDev& L) { return ((size_t)L.lds_tw + L.lds_bw + L.lds_sw + (L.lds_bw + 1) / 2) * 8; }
int build_contig_compact(mirge_ctx* c, Lib& L) {
    const uint64_t nbk = 1ull << (2 * L.kb);
    // an occupancy bitmap in front of the slots pays where most buckets are empty (it is 1/64 of the slots' bytes)
    L.ctab = TabDesc{(uint32_t)L.n_slots, (uint32_t)L.n_bitw, tab_info(L.kb, 0, 0, nbk >= 8 * L.T)};
    int rc = build_both(c, L, L.kb, 0, 0u, true);
    if (rc) return rc;
    L.lds_tw = (uint32_t)(L.T / 32 + 3); L.lds_bw = (uint32_t)(L.T / 64 + 2); L.lds_sw = L.has_n ? L.lds_bw : 0;
    // the reference of a position = the separators before it: per bitmap word a running count, the rest is a popcount
    std::vector<uint32_t> rank(L.lds_bw, 0);
    uint32_t run = 0;
    for (uint32_t j = 0; j < L.lds_bw; ++j) { rank[j] = run; run += (uint32_t)__builtin_popcountll(j < L.h_sep.size() ? L.h_sep[j] : 0ull); }
    HIPCHK(c, L.rank.ensure((size_t)L.lds_bw * 4, false, 0, c->st));
    HIPCHK(c, hipMemcpyAsync(L.rank.p, rank.data(), (size_t)L.lds_bw * 4, hipMemcpyHostToDevice, c->st));
    if (L.has_n) {
        HIPCHK(c, L.sep.ensure((size_t)L.lds_bw * 8, false, 0, c->st));
        HIPCHK(c, hipMemcpyAsync(L.sep.p, L.h_sep.data(), (size_t)L.lds_bw * 8, hipMemcpyHostToDevice, c->st));
    }
    HIPCHK(c, hipStreamSynchronize(c->st));                    // rank is a vector of this frame
    L.has_compact = true;
    return MIRGE_OK;
}
// Block scheme (blocks, m) of a library (kernels.h: block schemes), built the first time a cascade holds a round that can use it.
int ensure_scheme(mirge_ctx* c, Lib& L, uint32_t blocks, int m) {
    if (!L.loaded || !(L.has_both || L.has_compact) || L.T == 0) return MIRGE_OK;
    for (auto& x : L.sch) if (tab_blocks(x) == blocks && tab_m(x) == m) return MIRGE_OK;
    if ((int)L.sch.size() >= MAX_SCHEMES) return MIRGE_OK;
    const int ns = scheme_nshapes(m), kbs = std::min(scheme_max_sel(blocks, m), L.kb);
    const uint64_t nbk = 1ull << (2 * kbs);
    if (L.n_pent + (uint64_t)ns * 2 * L.T >= 0xFFFFFFF0ull || L.n_slots + ns * nbk >= 0xFFFFFFF0ull) return MIRGE_OK;    // too large a library: its short reads stay with the generic kernel
    if (L.has_compact && L.n_pent + (uint64_t)ns * 2 * L.T >= LDS_MAX_ENTRIES) return MIRGE_OK;
    const TabDesc t{(uint32_t)L.n_slots, (uint32_t)L.n_bitw, tab_info(kbs, blocks, m, L.has_compact ? nbk >= 8 * L.T : bits_wanted(nbk))};
    for (int i = 0; i < ns; ++i) { int rc = build_both(c, L, kbs, blocks, scheme_shape(m, i), L.has_compact); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->st));
    L.sch.push_back(t);
    return MIRGE_OK;
}
// the schemes a round can use: (4,4,4,4) for one mismatch (12-nt shapes); for two (5,5,5,5), (4,4,4,4), (3,3,3,3) (10 / 8 /
// 6-nt shapes) and, between the last two, windows of exactly 15 / 14 / 13 nt -- (4,4,4,3), (4,4,3,3), (4,3,3,3): 7-8 / 6-8 /
// 6-7-nt shapes -- for the reads that short (MIRGE_SCHEMES_UNEQ = 0: without these, A/B aid)
int ensure_schemes(mirge_ctx* c, Lib& L, const mirge_round& R) {
    if (const char* env = getenv("MIRGE_SCHEMES")) { if (atoi(env) == 0) return MIRGE_OK; }      // A/B aid
    bool uneq = true;
    if (const char* env = getenv("MIRGE_SCHEMES_UNEQ")) uneq = atoi(env) != 0;
    const int ms = std::min(R.seedmm, R.maxmm);
    int rc = MIRGE_OK;
    auto want = [&](int m) {
        if (m == 1 && !rc) rc = ensure_scheme(c, L, blocks_eq(4), 1);
        if (m == 2) {
            static const uint32_t bl[6] = {0x5555u, 0x4444u, 0x3444u, 0x3344u, 0x3334u, 0x3333u};
            for (int j = 0; j < 6 && !rc; ++j) if (uneq || bl[j] == blocks_eq(blk(bl[j], 0))) rc = ensure_scheme(c, L, bl[j], 2);
        }
    };
    want(ms);
    if (R.maxmm != ms && R.seedlen < 20) want(R.maxmm);         // windows that reach beyond a short seed fall under the whole-read allowance
    return rc;
}

// ---- library: host packing + device index build
int load_library(mirge_ctx* c, int lib, std::vector<std::string>&& names, const char* seqs, const uint64_t* offs) {
    if (lib < 0 || lib >= MIRGE_N_LIBS) return fail(c, MIRGE_E_ARG, "library index out of range");
    Lib& L = c->libs[lib];
    size_t n = names.size();
    uint64_t T = 0;
    for (size_t i = 0; i < n; ++i) T += (offs[i + 1] - offs[i]) + 1;
    if (T >= (1ull << 31)) return fail(c, MIRGE_E_LIMIT, "library larger than 2^31 nt");
    size_t words = (size_t)(T / 32) + 10, bwords = (size_t)(T / 64) + 10;
    std::vector<uint64_t> text(words, 0), bad(bwords, 0);
    L.h_sep.assign(bwords, 0); L.has_n = false;
    L.ref_start.assign(n + 1, 0);
    uint64_t p = 0;
    for (size_t i = 0; i < n; ++i) {
        L.ref_start[i] = (uint32_t)p;
        for (uint64_t j = offs[i]; j < offs[i + 1]; ++j, ++p) {
            uint8_t ch = (uint8_t)seqs[j];
            if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);
            int code = base_code(ch);
            if (code >= 0 && code < 4) text[p >> 5] |= (uint64_t)code << (2 * (p & 31));
            else { bad[p >> 6] |= 1ull << (p & 63); L.has_n = true; }      // reference N: not alignable
        }
        bad[p >> 6] |= 1ull << (p & 63);                           // separator
        L.h_sep[p >> 6] |= 1ull << (p & 63);
        ++p;
    }
    L.ref_start[n] = (uint32_t)p;
    L.T = T;
    L.names = std::move(names);
    L.short_names.clear();
    for (auto& nm : L.names) { size_t e = nm.find_first_of(" \t"); L.short_names.push_back(e == std::string::npos ? nm : nm.substr(0, e)); }
    int kb = (int)std::lround(std::log2((double)std::max<uint64_t>(T, 16)) / 2.0) + 1;   // ~0.25 random entries per bucket
    L.kb = std::min(14, std::max(4, kb));
    L.use_bits = true; L.has_both = false; L.has_compact = false; L.sch.clear(); L.n_slots = L.n_bitw = L.n_pent = 0;
    int kb_max = 14;
    if (const char* env = getenv("MIRGE_KB_MAX")) kb_max = std::min(15, std::max(4, atoi(env)));         // tuning aid
    L.kb = std::min(kb_max, std::max(4, kb));
    if (const char* env = getenv("MIRGE_KB_BIAS")) L.kb = std::min(kb_max, std::max(4, L.kb + atoi(env)));   // tuning aid
    if (const char* env = getenv("MIRGE_KB")) { if (atoi(env) >= 2 && atoi(env) <= 15) L.kb = atoi(env); }   // tests
    uint64_t nb = 1ull << (2 * L.kb);
    {
        uint64_t lim = 8ull << 20;                                    // koff bytes that stay cache-resident (MIRGE_BITS_ABOVE)
        if (const char* env = getenv("MIRGE_BITS_ABOVE")) lim = strtoull(env, nullptr, 10);
        L.use_bits = nb * 4 > lim;
    }
    HIPCHK(c, L.text.ensure(words * 8, false, 0, c->st));
    HIPCHK(c, L.bad.ensure(bwords * 8, false, 0, c->st));
    HIPCHK(c, L.dref.ensure((n + 1) * 4, false, 0, c->st));
    HIPCHK(c, L.koff.ensure((nb + 1) * 4, false, 0, c->st));
    HIPCHK(c, L.kent.ensure(std::max<uint64_t>(T, 1) * sizeof(KEnt), false, 0, c->st));
    HIPCHK(c, L.kpre.ensure(std::max<uint64_t>(T, 1) * 8, false, 0, c->st));
    HIPCHK(c, c->hist.ensure((nb + 1) * 4, false, 0, c->st));
    HIPCHK(c, L.kbits.ensure((nb / 32 + 1) * 4, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(L.kbits.p, 0, (nb / 32 + 1) * 4, c->st));
    HIPCHK(c, hipMemcpyAsync(L.text.p, text.data(), words * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(L.bad.p, bad.data(), bwords * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(L.dref.p, L.ref_start.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->st));
    std::vector<uint32_t> lut;
    L.lut_shift = ref_lut_shift(T);
    build_ref_lut(L.ref_start.data(), (uint32_t)n, T, L.lut_shift, lut);
    HIPCHK(c, L.dlut.ensure(lut.size() * 4, false, 0, c->st));
    HIPCHK(c, hipMemcpyAsync(L.dlut.p, lut.data(), lut.size() * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemsetAsync(c->hist.p, 0, (nb + 1) * 4, c->st));
    if (T > 0) {
        uint32_t grid = (uint32_t)((T + 255) / 256);
        hipLaunchKernelGGL(k_kmer_hist, dim3(grid), dim3(256), 0, c->st, L.text.as<uint64_t>(), (uint32_t)T, L.kb, c->hist.as<uint32_t>());
        hipLaunchKernelGGL(k_bucket_bits, dim3((uint32_t)((nb / 32 + 256) / 256)), dim3(256), 0, c->st, c->hist.as<uint32_t>(), nb, L.kbits.as<uint32_t>());
        int rc = scan_exclusive(c, c->hist.as<uint32_t>(), L.koff.as<uint32_t>(), nb + 1);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->hist.p, L.koff.p, nb * 4, hipMemcpyDeviceToDevice, c->st));
        hipLaunchKernelGGL(k_kmer_scatter, dim3(grid), dim3(256), 0, c->st, L.text.as<uint64_t>(), (uint32_t)T, L.kb, c->hist.as<uint32_t>(), L.kent.as<KEnt>(), L.kpre.as<uint64_t>());
        HIPCHK(c, hipGetLastError());
        int min_kb = 9;
        if (const char* env = getenv("MIRGE_SLOTS_MIN_KB")) min_kb = atoi(env);
        if (want_slots() && lds_wanted(L)) { rc = build_contig_compact(c, L); if (rc) return rc; }
        else if (L.kb >= min_kb && want_slots()) { rc = build_contig_both(c, L); if (rc) return rc; }
    } else {
        HIPCHK(c, hipMemsetAsync(L.koff.p, 0, (nb + 1) * 4, c->st));
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    L.loaded = true;
    c->annotated = false; c->counts_valid = false; c->host_valid = false;
    return MIRGE_OK;
}

int read_file(mirge_ctx* c, const char* path, std::string& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, MIRGE_E_IO, std::string("cannot open ") + path);
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)std::max(0L, sz));
    size_t got = sz > 0 ? fread(&out[0], 1, (size_t)sz, f) : 0;
    fclose(f);
    if ((long)got != sz) return fail(c, MIRGE_E_IO, std::string("short read on ") + path);
    return MIRGE_OK;
}

int load_fasta(mirge_ctx* c, int lib, const char* path) {
    std::string data;
    int rc = read_file(c, path, data);
    if (rc) return rc;
    std::vector<std::string> names; std::string seqs; std::vector<uint64_t> offs;
    size_t i = 0, n = data.size();
    while (i < n) {
        size_t e = data.find('\n', i); if (e == std::string::npos) e = n;
        size_t le = e; while (le > i && (data[le - 1] == '\r')) --le;
        if (le > i && data[i] == '>') { names.emplace_back(data, i + 1, le - i - 1); offs.push_back(seqs.size()); }
        else if (!names.empty()) seqs.append(data, i, le - i);
        else if (le > i) return fail(c, MIRGE_E_FORMAT, std::string(path) + ": sequence data before the first '>' header");
        i = e + 1;
    }
    offs.push_back(seqs.size());
    return load_library(c, lib, std::move(names), seqs.data(), offs.data());
}

// a library out of a bowtie 1 index (ebwt.h: layout unvalidated, every implied size cross-checked)
int load_ebwt(mirge_ctx* c, int lib, const std::string& base) {
    std::vector<std::string> names; std::string seqs; std::vector<uint64_t> offs; std::string err;
    if (ebwt_extract(base, names, seqs, offs, err) != 0) return fail(c, MIRGE_E_FORMAT, err);
    return load_library(c, lib, std::move(names), seqs.data(), offs.data());
}

// ---- collapse of one key class, in two halves so that the two classes overlap: everything up to the emit kernel is
// queued on the class's stream by collapse_launch; collapse_finish waits for it, learns how many uniques were new
// and appends the sample's segment.
// the partitioned form (kernels.h: k_part_pass / k_part_fold) takes inputs its fixed fan-out suits: from MIRGE_PART_MIN
// keys (default 2^20: below that the table form's launches are already latency-bound) up to 2^25 (8192 final partitions
// of <= 4096 records); beyond, and whenever a region overruns, the table form
// several small regions zeroed by ONE launch: a hipMemsetAsync is a launch of its own (two, when the size is not a
// multiple of its fill width), ~10 us apart at the pace the host can queue them -- the step spent 45 us on four of them
// before its first kernel
struct ZeroJob { unsigned long long* p[4]; uint32_t n[4]; };      // 8-byte words
__global__ void __launch_bounds__(256) k_zero_regions(ZeroJob J) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < J.n[r]; i += gridDim.x * 256) J.p[r][i] = 0ull;
}
inline void zero_regions(hipStream_t st, unsigned long long* p0, size_t n0, unsigned long long* p1 = nullptr, size_t n1 = 0,
                         unsigned long long* p2 = nullptr, size_t n2 = 0, unsigned long long* p3 = nullptr, size_t n3 = 0) {
    ZeroJob J{{p0, p1, p2, p3}, {(uint32_t)n0, (uint32_t)n1, (uint32_t)n2, (uint32_t)n3}};
    const size_t mx = std::max(std::max(n0, n1), std::max(n2, n3));
    hipLaunchKernelGGL(k_zero_regions, dim3((uint32_t)std::max<size_t>(1, std::min<size_t>(512, (mx + 255) / 256))), dim3(256), 0, st, J);
}
bool part_wanted(uint64_t n) {
    uint64_t lo = 1ull << 20, hi = 176000000ull;
    if (const char* e = getenv("MIRGE_PART_MIN")) lo = strtoull(e, nullptr, 10);
    if (const char* e = getenv("MIRGE_PART_MAX")) hi = strtoull(e, nullptr, 10);
    return n >= lo && n <= hi;
}
// fan-out of the first pass: the fewest final partitions (P1 x PT_P2 of them, PT_C2 records each at most) that come out
// at most ~75 % full when every read is distinct (hashed keys: 2048 is then 13 sigma above the mean).  Fewest, because
// every partition ends in one atomic on the table's rank counter and those are served one after the other (~12 ns):
// at 100M reads the fold kernel took 1.68 ms with 131 K partitions -- exactly that chain.
inline uint64_t part_fill() { if (const char* e = getenv("MIRGE_PART_FILL")) { const long v = atol(e); if (v >= 100000 && v <= 500000) return (uint64_t)v; } return 390000; }     // reads per first-pass bucket (tuning aid)
int part_p1(uint64_t n) { return (int)std::min<uint64_t>(512, std::max<uint64_t>(16, (n + part_fill() - 1) / part_fill())); }
// cursor groups of the first split pass (kernels.h: k_part_pass says what they are for)
inline uint32_t part_groups(uint64_t n) {
    if (const char* e = getenv("MIRGE_PART_GROUPS")) { const long v = atol(e); if (v >= 1 && v <= 64) return (uint32_t)v; }
    return 8;
}
// 8-byte words between two cursors of the first pass: every tile of the launch bumps every one of them, and the device
// serves atomics on one 128-B LINE one after the other (profiles/r03/collapse_experiments.txt: 16 cursors per line 0.269 ms,
// a line each 0.235, 512 B apart 0.224)
inline uint32_t part_cstride() {
    if (const char* e = getenv("MIRGE_PART_CSTRIDE")) { const long v = atol(e); if (v >= 1 && v <= 4096) return (uint32_t)v; }
    return 64;
}
template <int KW>
int collapse_launch(mirge_ctx* c, Class& C, const uint64_t* d_keys, uint64_t n64, int sample, const char* tagA, const char* tagB,
                    const uint32_t* d_wt = nullptr, bool allow_part = true) {
    if (n64 >= 0xFFFFFFF0ull) return fail(c, MIRGE_E_LIMIT, "more than 2^32 reads in one collapse call");
    if (C.U + n64 >= 0xFFFFFFF0ull) return fail(c, MIRGE_E_LIMIT, "more than 2^32 unique sequences");
    uint32_t n = (uint32_t)n64;
    C.pending = true; C.pend_n = n; C.pend_sample = sample; C.pend_part = false; C.pend_keys = d_keys; C.pend_wt = d_wt;
    if (n == 0) return MIRGE_OK;
    hipStream_t st = C.st;
    if (KW == 2 && allow_part && part_wanted(n)) {
        const int PT_P1 = part_p1(n);
        const uint32_t G = part_groups(n);                                   // cursor groups of the first pass (kernels.h: k_part_pass)
        const uint32_t PG = (uint32_t)PT_P1 * G;                             // regions the first pass writes = input partitions of the second
        uint32_t C1 = n / PG + n / (2 * PG) + PT_TILE2;                      // 1.5 x the even share
        if (const char* e = getenv("MIRGE_PART_C1")) C1 = (uint32_t)std::max(1, atoi(e));       // tests: force an overrun
        const uint32_t CS = part_cstride();                                  // 8-byte words between two cursors of the first pass
        const uint32_t CS2 = 1;                                              // ... of the second pass: dense (16 to a line: its 256 atomics per tile touch fewer lines; a line each 0.216 against 0.193 ms)
        const size_t ncur = (size_t)PG * CS + (size_t)PT_P1 * PT_P2 * CS2 + 1;
        HIPCHK(c, C.pt_k1.ensure((size_t)PG * C1 * 16, false, 0, st));
        HIPCHK(c, C.pt_w1.ensure((size_t)PG * C1 * 4, false, 0, st));
        HIPCHK(c, C.pt_k2.ensure((size_t)PT_P1 * PT_P2 * PT_C2 * 16, false, 0, st));
        HIPCHK(c, C.pt_w2.ensure((size_t)PT_P1 * PT_P2 * PT_C2 * 4, false, 0, st));
        HIPCHK(c, C.pt_cur.ensure(ncur * 8, false, 0, st));
        HIPCHK(c, C.ws_ctr.ensure(64, false, 0, st));
        HIPCHK(c, C.ukeys.ensure((C.U + n) * KW * 8, true, C.U * KW * 8, st));
        HIPCHK(c, C.coo_g.ensure((C.coo_n + n) * 4, true, C.coo_n * 4, st));
        HIPCHK(c, C.coo_c.ensure((C.coo_n + n) * 4, true, C.coo_n * 4, st));
        const bool use_g = C.U > 0;
        if (use_g) {
            uint64_t need = pow2_at_least(2ull * (C.U + n));
            if (need > (1ull << 32)) need = 1ull << 32;
            if (!C.gvalid || C.gcap < need) {
                HIPCHK(c, C.gslots.ensure(need * 4, false, 0, st));
                C.gcap = need;
                HIPCHK(c, hipMemsetAsync(C.gslots.p, 0xFF, need * 4, st));
                hipLaunchKernelGGL((k_gtable_insert<2>), dim3((uint32_t)((C.U + 255) / 256)), dim3(256), 0, st,
                                   C.ukeys.as<uint64_t>(), 0u, (uint32_t)C.U, C.gslots.as<uint32_t>(), (uint32_t)(C.gcap - 1));
                C.gvalid = true;
            }
        }
        bool zero_ll = false;
        if (C.ll_ok) {
            for (auto& b : C.ll) HIPCHK(c, b.ensure((C.U + n) * 4, true, C.U * 4, st));
            HIPCHK(c, C.ll_ctr.ensure(LL_CSTRIDE * 16, false, 0, st));
            if (!C.ll_fresh) { zero_ll = true; C.ll_fresh = true; }
        }
        // cursors, the rank counter (+ the overflow flag beside it: word 2, so that ONE copy brings both to the host) and
        // the length-class lists' length words
        zero_regions(st, C.pt_cur.as<unsigned long long>(), ncur, C.ws_ctr.as<unsigned long long>(), 8,
                     zero_ll ? C.ll_ctr.as<unsigned long long>() : nullptr, zero_ll ? (size_t)LL_CSTRIDE * 2 : 0);
        unsigned long long* cur = C.pt_cur.as<unsigned long long>();
        unsigned long long* ovf = C.ws_ctr.as<unsigned long long>() + 2;
        kt_begin(c, "collapse_part16", st);
#define MG_PASS1(NB) k_part_pass<NB, PT_TILE1, PT_BLOCK1>
        auto pass1 = PT_P1 <= 16 ? MG_PASS1(16) : PT_P1 <= 32 ? MG_PASS1(32) : PT_P1 <= 64 ? MG_PASS1(64) : PT_P1 <= 128 ? MG_PASS1(128) : PT_P1 <= 256 ? MG_PASS1(256) : MG_PASS1(512);      // (NB = LDS counters: any fan-out up to it)
#undef MG_PASS1
        const uint32_t g1 = (n + PT_TILE1 - 1) / PT_TILE1, g2 = (C1 + PT_TILE2 - 1) / PT_TILE2;      // a block per tile
        hipLaunchKernelGGL(pass1, dim3(g1, 1), dim3(PT_BLOCK1), 0, st, d_keys, d_wt,
                           (const unsigned long long*)nullptr, n, 0u, (uint32_t)PT_P1, 0, C.pt_k1.as<uint64_t>(), C.pt_w1.as<uint32_t>(), cur, C1, ovf, G, 1u, 1u, CS);
        hipLaunchKernelGGL((k_part_pass<PT_P2, PT_TILE2, PT_BLOCK2>), dim3(g2, PG), dim3(PT_BLOCK2), 0, st, C.pt_k1.as<uint64_t>(),
                           C.pt_w1.as<uint32_t>(), (const unsigned long long*)cur, C1, C1, (uint32_t)PT_P1, 1, C.pt_k2.as<uint64_t>(), C.pt_w2.as<uint32_t>(),
                           cur + (size_t)PG * CS, (uint32_t)PT_C2, ovf, 1u, G, CS, CS2);
        kt_end(c, "collapse_part16", (uint64_t)n * 16, st);
        kt_begin(c, "collapse_fold16", st);
        hipLaunchKernelGGL(k_part_fold, dim3(PT_P1 * PT_P2), dim3(PT_FBLOCK), 0, st, C.pt_k2.as<uint64_t>(), C.pt_w2.as<uint32_t>(),
                           (const unsigned long long*)(cur + (size_t)PG * CS), use_g ? C.gslots.as<uint32_t>() : (const uint32_t*)nullptr,
                           (uint32_t)(C.gcap ? C.gcap - 1 : 0), C.ukeys.as<uint64_t>(), (uint32_t)C.U,
                           C.coo_g.as<uint32_t>() + C.coo_n, C.coo_c.as<uint32_t>() + C.coo_n, C.ws_ctr.as<unsigned long long>(),
                           C.ll_ok ? C.ll[0].as<uint32_t>() : (uint32_t*)nullptr, C.ll[1].as<uint32_t>(), C.ll_ctr.as<unsigned long long>(), LL_SPLIT, CS2);
        kt_end(c, "collapse_fold16", 0, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(C.h_ctr, C.ws_ctr.p, 24, hipMemcpyDeviceToHost, st));      // rank counter words + the overflow flag
        C.pend_part = true;
        return MIRGE_OK;
    }
    C.ll_ok = false;                                         // the table form keeps no length-class lists
    uint64_t capL = pow2_at_least((uint64_t)n + n / 4);      // load <= 0.8 even if every read is distinct; smaller table = more of it cache-resident
    if (capL > (1ull << 32)) {                               // slots are addressed by 32 hash bits
        if ((uint64_t)n > 3ull << 30) return fail(c, MIRGE_E_LIMIT, "more than 3 * 2^30 reads in one collapse call");
        capL = 1ull << 32;
    }
    HIPCHK(c, C.ws_lslots.ensure(capL * 4, false, 0, st));
    HIPCHK(c, C.ws_cnt.ensure((size_t)n * 4, false, 0, st));
    HIPCHK(c, C.ws_cnt0.ensure((size_t)n * 4, false, 0, st));
    HIPCHK(c, C.ws_ctr.ensure(64, false, 0, st));
    HIPCHK(c, C.ukeys.ensure((C.U + n) * KW * 8, true, C.U * KW * 8, st));
    HIPCHK(c, C.coo_g.ensure((C.coo_n + n) * 4, true, C.coo_n * 4, st));
    HIPCHK(c, C.coo_c.ensure((C.coo_n + n) * 4, true, C.coo_n * 4, st));
    bool use_g = C.U > 0;
    if (use_g) {
        uint64_t need = pow2_at_least(2ull * (C.U + n));
        if (need > (1ull << 32)) {
            if (C.U + n > 3ull << 30) return fail(c, MIRGE_E_LIMIT, "more than 3 * 2^30 unique sequences");
            need = 1ull << 32;
        }
        if (!C.gvalid || C.gcap < need) {
            HIPCHK(c, C.gslots.ensure(need * 4, false, 0, st));
            C.gcap = need;
            HIPCHK(c, hipMemsetAsync(C.gslots.p, 0xFF, need * 4, st));
            hipLaunchKernelGGL((k_gtable_insert<KW>), dim3((uint32_t)((C.U + 255) / 256)), dim3(256), 0, st,
                               C.ukeys.as<uint64_t>(), 0u, (uint32_t)C.U, C.gslots.as<uint32_t>(), (uint32_t)(C.gcap - 1));
            C.gvalid = true;
        }
    }
    HIPCHK(c, hipMemsetAsync(C.ws_lslots.p, 0xFF, capL * 4, st));
    HIPCHK(c, hipMemsetAsync(C.ws_cnt.p, 0, (size_t)n * 4, st));
    HIPCHK(c, hipMemsetAsync(C.ws_ctr.p, 0, 64, st));
#ifndef MG_CL_T
#define MG_CL_T 2048
#endif
#ifndef MG_CL_B
#define MG_CL_B 1024
#endif
    constexpr int T = KW == 2 ? MG_CL_T : 512;
    constexpr int CB = KW == 2 ? MG_CL_B : 256;
    kt_begin(c, tagA, st);
    hipLaunchKernelGGL((k_collapse_local<KW, T, CB>), dim3((n + T - 1) / T), dim3(CB), 0, st,
                       d_keys, d_wt, n, C.ws_lslots.as<uint32_t>(), (uint32_t)(capL - 1), C.ws_cnt.as<uint32_t>(), C.ws_cnt0.as<uint32_t>());
    kt_end(c, tagA, (uint64_t)n * KW * 8, st);
    kt_begin(c, tagB, st);
    hipLaunchKernelGGL((k_collapse_emit<KW>), dim3((n + 256 * EMIT_PER - 1) / (256 * EMIT_PER)), dim3(256), 0, st,
                       d_keys, C.ws_cnt.as<uint32_t>(), C.ws_cnt0.as<uint32_t>(), n, use_g ? C.gslots.as<uint32_t>() : (const uint32_t*)nullptr,
                       (uint32_t)(C.gcap ? C.gcap - 1 : 0), C.ukeys.as<uint64_t>(), (uint32_t)C.U,
                       C.coo_g.as<uint32_t>() + C.coo_n, C.coo_c.as<uint32_t>() + C.coo_n,
                       C.ws_ctr.as<unsigned long long>());
    kt_end(c, tagB, 0, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(C.h_ctr, C.ws_ctr.p, 16, hipMemcpyDeviceToHost, st));
    return MIRGE_OK;
}
template <int KW>
int collapse_finish(mirge_ctx* c, Class& C, const char* tagB) {
    if (!C.pending) return MIRGE_OK;
    C.pending = false;
    if (C.pend_n == 0) { C.segs.push_back({C.pend_sample, C.coo_n, 0}); return MIRGE_OK; }
    HIPCHK(c, hipStreamSynchronize(C.st));
    if (C.pend_part && C.h_ctr[2]) {
        // a region of the partitioned form overran (skewed hash bits, or more distinct records than its fan-out holds):
        // nothing it wrote is kept -- the table form takes the same input
        ktime(c, "collapse_part16").used = false; ktime(c, "collapse_fold16").used = false;
        int rc = collapse_launch<KW>(c, C, C.pend_keys, C.pend_n, C.pend_sample, "collapse_local16", "collapse_emit16", C.pend_wt, false);
        C.pending = false;
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(C.st));
    }
    uint64_t n_new = C.h_ctr[0] >> 32, n_emit = C.h_ctr[0] & 0xFFFFFFFFull;       // k_collapse_emit / k_part_fold pack both into one word
    ktime(c, C.pend_part ? "collapse_fold16" : tagB).bytes = n_new * (KW * 8 + 4) + n_emit * 4;
    if (C.gvalid && n_new) {
        hipLaunchKernelGGL((k_gtable_insert<KW>), dim3((uint32_t)((n_new + 255) / 256)), dim3(256), 0, C.st,
                           C.ukeys.as<uint64_t>(), (uint32_t)C.U, (uint32_t)(C.U + n_new), C.gslots.as<uint32_t>(), (uint32_t)(C.gcap - 1));
        HIPCHK(c, hipGetLastError());
    }
    C.segs.push_back({C.pend_sample, C.coo_n, n_emit});
    C.U += n_new; C.coo_n += n_emit;
    return MIRGE_OK;
}

// both classes of one sample: the 16-byte class on st, the 48-byte class on st2, side by side.
// defer: everything is queued and nothing is waited for -- the unique counts stay on the device (Class::pending), the
// cascade is queued behind (annotate() sizes its launches by upper bounds and reads the counts there), and
// collapse_settle() does the bookkeeping once the host has waited for the cascade anyway.
int collapse_both(mirge_ctx* c, const uint64_t* d16, uint64_t n16, const uint32_t* w16, const uint64_t* d48, uint64_t n48, const uint32_t* w48, int sample,
                  bool defer = false) {
    int rc = fork_st2(c);                                  // the inputs were produced on st
    if (!rc) rc = collapse_launch<2>(c, c->cls[0], d16, n16, sample, "collapse_local16", "collapse_emit16", w16);
    if (!rc) rc = collapse_launch<6>(c, c->cls[1], d48, n48, sample, "collapse_local48", "collapse_emit48", w48);
    if (rc) {                                              // a failed launch: drop whatever was queued
        c->cls[0].pending = c->cls[1].pending = false;
        (void)hipStreamSynchronize(c->st); (void)hipStreamSynchronize(c->st2);
        return rc;
    }
    if (defer) return join_st2(c);
    rc = collapse_finish<2>(c, c->cls[0], "collapse_emit16");
    int r1 = collapse_finish<6>(c, c->cls[1], "collapse_emit48");
    if (!rc) rc = r1;
    if (!rc) rc = join_st2(c);
    return rc;
}
// a deferred collapse whose kernels are known to have finished: the two halves' bookkeeping.  reran: a partitioned
// collapse had overrun and the sample went through the table form just now (whatever was computed from the first
// attempt's output is void).
int collapse_settle(mirge_ctx* c, bool& reran) {
    reran = c->cls[0].pending && c->cls[0].pend_n && c->cls[0].pend_part && c->cls[0].h_ctr[2];
    int rc = collapse_finish<2>(c, c->cls[0], "collapse_emit16");
    int r1 = collapse_finish<6>(c, c->cls[1], "collapse_emit48");
    if (!rc) rc = r1;
    float ms = 0; if (c->timing) (void)hipEventElapsedTime(&ms, c->ev_c[0], c->ev_c[1]); c->collapse_ms = ms;
    return rc;
}

int collapse_packed(mirge_ctx* c, const void* d16, uint64_t n16, const void* d48, uint64_t n48, int sample, bool defer = false) {
    if (sample < 0 || sample >= c->S) return fail(c, MIRGE_E_ARG, "sample index out of range");
    c->annotated = false; c->counts_valid = false; c->host_valid = false;
    kt_clear(c, "collapse");
    hipEvent_t a = c->ev_c[0], b = c->ev_c[1];
    if (c->timing) HIPCHK(c, hipEventRecord(a, c->st));
    int rc = collapse_both(c, (const uint64_t*)d16, n16, nullptr, (const uint64_t*)d48, n48, nullptr, sample, defer);
    if (!rc) {
        if (c->timing) HIPCHK(c, hipEventRecord(b, c->st));
        if (defer) return MIRGE_OK;
        HIPCHK(c, hipStreamSynchronize(c->st));
        float ms = 0; if (c->timing) (void)hipEventElapsedTime(&ms, a, b); c->collapse_ms = ms;
        kt_resolve(c);
    }
    return rc;
}

// pack a device-resident ASCII matrix into the two key streams at out16/out48 (appending at the
// positions held in c->ctr[0..1]); reports kernel-detected input errors.
int pack_launch(mirge_ctx* c, const void* d_seqs, uint32_t width, const void* d_lens, uint64_t n, void* out16, void* out48) {
    if (width == 0 || width % 16) return fail(c, MIRGE_E_ARG, "matrix width must be a positive multiple of 16");
    if (n == 0) return MIRGE_OK;
    kt_begin(c, "pack");
    hipLaunchKernelGGL(k_pack, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->st, (const uint8_t*)d_seqs, width,
                       (const uint8_t*)d_lens, n, (uint64_t*)out16, (uint64_t*)out48, c->ctr.as<unsigned long long>());
    kt_end(c, "pack", n * 16);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->ctr.p, 32, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (c->h_ctr[2] == 4) return fail(c, MIRGE_E_FORMAT, "row " + std::to_string(c->h_ctr[3]) + " of the ASCII matrix holds a character other than A, C, G, T, N (the matrix form takes reads a key can hold)");
    if (c->h_ctr[2] == 5) return fail(c, MIRGE_E_LIMIT, "row " + std::to_string(c->h_ctr[3]) + " of the ASCII matrix is longer than " + std::to_string(KeyTraits<6>::MAXLEN) + " nt (the matrix form takes reads a key can hold)");
    return MIRGE_OK;
}

// 0 = a key holds the read, 1 = a verbatim read, 4 / 5 = refused (control character or blank / longer than MIRGE_MAX_READ_LEN)
inline int read_kind(const char* s, uint64_t l) {
    if (l > MIRGE_MAX_READ_LEN) return 5;
    bool plain = l <= (uint64_t)KeyTraits<6>::MAXLEN;
    for (uint64_t i = 0; i < l; ++i) {
        const uint8_t ch = (uint8_t)s[i];
        if (ch <= 0x20 || ch >= 0x7F) return 4;
        plain = plain && base_code(ch) >= 0;
    }
    return plain ? 0 : 1;
}
int host_threads() {
    int n = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("MIRGE_HOST_THREADS")) n = atoi(e);
    return std::max(1, std::min(n, 16));
}
// fn(lo, hi, t) over [0, n) cut into one slice per thread
template <class F> void parallel_slices(uint64_t n, int nthr, F fn) {
    if (nthr <= 1 || n < 4096) { fn((uint64_t)0, n, 0); return; }
    std::vector<std::thread> thr;
    for (int t = 0; t < nthr; ++t) thr.emplace_back([&, t] { fn(n * t / nthr, n * (t + 1) / nthr, t); });
    for (auto& x : thr) x.join();
}
int verb_refused(mirge_ctx* c, int kind, uint64_t read) {
    if (kind == 4) return fail(c, MIRGE_E_FORMAT, "read " + std::to_string(read) + " holds a control character, a blank or a byte outside ASCII (a CR line end among them)");
    return fail(c, MIRGE_E_LIMIT, "read " + std::to_string(read) + " is longer than " + std::to_string(MIRGE_MAX_READ_LEN) + " nt");
}

// host reads (concatenated + offsets) -> chunks of ASCII matrix -> device pack -> collapse.  The staging (classify every
// read, lay the ones a key holds out as matrix rows) runs on the host's threads; reads no key holds go to the verbatim table.
int collapse_host(mirge_ctx* c, const char* seqs, const uint64_t* offs, uint64_t n, int sample) {
    if (sample < 0 || sample >= c->S) return fail(c, MIRGE_E_ARG, "sample index out of range");
    const int nthr = host_threads();
    const uint64_t CH = 1u << 22;
    HIPCHK(c, c->ctr.ensure(128, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->ctr.p, 0, 128, c->st));
    HIPCHK(c, c->stage16.ensure(std::max<uint64_t>(n, 1) * 16, false, 0, c->st));
    std::vector<uint8_t> mat, lens, kind;
    std::vector<uint64_t> verb_later;                            // reads no key holds, by index
    uint64_t n16 = 0, n48 = 0;
    for (uint64_t b = 0; b < n; b += CH) {
        const uint64_t m = std::min(CH, n - b);
        kind.resize((size_t)m);
        std::vector<uint64_t> t_plain((size_t)nthr + 1, 0), t_max((size_t)nthr, 0);
        parallel_slices(m, nthr, [&](uint64_t lo, uint64_t hi, int t) {
            uint64_t np = 0, mx = 0;
            for (uint64_t i = lo; i < hi; ++i) {
                const uint64_t l = offs[b + i + 1] - offs[b + i];
                const int k = read_kind(seqs + offs[b + i], l);
                kind[(size_t)i] = (uint8_t)k;
                if (k == 0) { ++np; mx = std::max(mx, l); }
            }
            t_plain[(size_t)t + 1] = np; t_max[(size_t)t] = mx;
        });
        for (uint64_t i = 0; i < m; ++i) if (kind[(size_t)i] > 1) return verb_refused(c, kind[(size_t)i], b + i);
        uint64_t maxlen = 0;
        for (int t = 0; t < nthr; ++t) { t_plain[(size_t)t + 1] += t_plain[(size_t)t]; maxlen = std::max(maxlen, t_max[(size_t)t]); }
        const uint64_t rows = t_plain[(size_t)nthr];
        const uint32_t W = (uint32_t)std::max<uint64_t>(16, (maxlen + 15) / 16 * 16);
        if (rows) {
            mat.assign((size_t)rows * W, 0); lens.resize((size_t)rows);
            parallel_slices(m, nthr, [&](uint64_t lo, uint64_t hi, int t) {
                uint64_t r = m < 4096 || nthr <= 1 ? 0 : t_plain[(size_t)t];
                for (uint64_t i = lo; i < hi; ++i) {
                    if (kind[(size_t)i]) continue;
                    const uint64_t l = offs[b + i + 1] - offs[b + i];
                    memcpy(&mat[(size_t)r * W], seqs + offs[b + i], (size_t)l);
                    lens[(size_t)r] = (uint8_t)l; ++r;
                }
            });
            HIPCHK(c, c->in_seqs.ensure((size_t)rows * W, false, 0, c->st));
            HIPCHK(c, c->in_lens.ensure((size_t)rows, false, 0, c->st));
            HIPCHK(c, c->stage48.ensure((n48 + rows) * 48, true, n48 * 48, c->st));
            HIPCHK(c, hipMemcpyAsync(c->in_seqs.p, mat.data(), (size_t)rows * W, hipMemcpyHostToDevice, c->st));
            HIPCHK(c, hipMemcpyAsync(c->in_lens.p, lens.data(), (size_t)rows, hipMemcpyHostToDevice, c->st));
            int rc = pack_launch(c, c->in_seqs.p, W, c->in_lens.p, rows, c->stage16.p, c->stage48.p);
            if (rc) return rc;
            n16 = c->h_ctr[0]; n48 = c->h_ctr[1];
        }
        if (rows < m)
            for (uint64_t i = 0; i < m; ++i)
                if (kind[(size_t)i] == 1) verb_later.push_back(b + i);
    }
    // the verbatim reads join their table only once nothing can fail any more: a refused read in a later chunk, or a
    // failed collapse, leaves the context as it was before the call
    int rc = collapse_packed(c, c->stage16.p, n16, c->stage48.p, n48, sample);
    if (rc) return rc;
    c->vb.add_batch(verb_later.size(), [&](uint64_t j) { const uint64_t i = verb_later[(size_t)j]; return std::make_pair(seqs + offs[i], (size_t)(offs[i + 1] - offs[i])); },
                    sample, c->S, nthr);
    return MIRGE_OK;
}


// ---- FASTQ ingest (reference miRge.pl:299-307).  Raw bytes in HBM -> record starts -> keys.
constexpr int FQ_SLOTS = 8;
constexpr size_t FQ_SLOT_BYTES = 16u << 20;
constexpr uint64_t FQ_WINDOW_DEFAULT = 1ull << 31;  // bytes per count/scan/index window: newline ordinals of a window fit uint32
uint64_t fq_window() {                              // MIRGE_FQ_WINDOW: smaller windows, so tests reach the multi-window path with small files
    if (const char* e = getenv("MIRGE_FQ_WINDOW")) {
        uint64_t w = strtoull(e, nullptr, 10) / FQ_SEG * FQ_SEG;
        if (w >= (uint64_t)FQ_SEG && w <= FQ_WINDOW_DEFAULT) return w;
    }
    return FQ_WINDOW_DEFAULT;
}

// The trimming step of a raw-FASTQ call (trim.h).  adapter: "none", "illumina", "ion" (miRge.pl:80-86), "+N" / "-N",
// or up to 16 comma-separated 3' adapter sequences of <= 64 nt ("name=SEQ", trailing "$" = anchored).  base is filled in per file (phred detection).
struct TrimPlan { bool on = false; TrimDev dev; uint64_t kept = 0; int phred_workers = 33, phred_reported = 33; };
int make_trim_plan(mirge_ctx* c, const char* adapter, int trim5, int trim3, TrimPlan& tp) {
    tp.on = true;
    std::string msg;
    int rc = trim_dev_from_spec(adapter, trim5, trim3, tp.dev, msg);
    return rc ? fail(c, rc == -5 ? MIRGE_E_LIMIT : MIRGE_E_ARG, msg) : MIRGE_OK;
}

// d: 16-byte aligned device pointer to `size` bytes; last = the file's last byte (decides whether the final line is terminated).
// tp (nullable): the file is RAW reads and goes through the trimming step; null: it is the trimmed file quantReads opens.
int collapse_fastq_dev(mirge_ctx* c, const uint8_t* d, uint64_t size, uint8_t last, int sample, uint64_t* n_reads, TrimPlan* tp = nullptr) {
    if (sample < 0 || sample >= c->S) return fail(c, MIRGE_E_ARG, "sample index out of range");
    if ((uintptr_t)d & 15) return fail(c, MIRGE_E_ARG, "FASTQ device buffer must be 16-byte aligned");
    kt_clear(c, "fastq");
    const uint64_t FQ_WINDOW = fq_window();
    HIPCHK(c, c->ctr.ensure(128, false, 0, c->st));
    uint64_t newlines = 0;
    std::vector<uint64_t> win_nl;
    const uint32_t max_seg = (uint32_t)(FQ_WINDOW / FQ_SEG);
    HIPCHK(c, c->fq_seg.ensure(((size_t)max_seg + 1) * 4, false, 0, c->st));
    HIPCHK(c, c->fq_segb.ensure(((size_t)max_seg + 1) * 4, false, 0, c->st));
    // pass 1 per window: newline count (the record count is needed before start[] can be sized)
    kt_begin(c, "fastq_count");
    for (uint64_t w0 = 0; w0 < size; w0 += FQ_WINDOW) {
        uint64_t wsz = std::min(FQ_WINDOW, size - w0);
        uint32_t nseg = (uint32_t)((wsz + FQ_SEG - 1) / FQ_SEG);
        HIPCHK(c, hipMemsetAsync(c->fq_seg.as<uint32_t>() + nseg, 0, 4, c->st));
        hipLaunchKernelGGL(k_fq_count, dim3(nseg), dim3(256), 0, c->st, d + w0, wsz, c->fq_seg.as<uint32_t>());
        int rc = scan_exclusive(c, c->fq_seg.as<uint32_t>(), c->fq_segb.as<uint32_t>(), (uint64_t)nseg + 1);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_ctr + 8, c->fq_segb.as<uint32_t>() + nseg, 4, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        win_nl.push_back((uint32_t)c->h_ctr[8]);
        newlines += win_nl.back();
    }
    kt_end(c, "fastq_count", size);
    uint64_t lines = newlines + ((size > 0 && last != '\n') ? 1 : 0);
    uint64_t n = (lines + 3) / 4;
    if (n_reads) *n_reads = n;
    if (n >= 0xFFFFFFF0ull) return fail(c, MIRGE_E_LIMIT, "more than 2^32 reads in one collapse call");
    HIPCHK(c, c->fq_start.ensure(std::max<uint64_t>(n, 1) * 8, false, 0, c->st));
    HIPCHK(c, c->stage16.ensure(std::max<uint64_t>(n, 1) * 16, false, 0, c->st));
    HIPCHK(c, c->stage48.ensure(std::max<uint64_t>(n, 1) * 48, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->fq_start.p, 0xFF, std::max<uint64_t>(n, 1) * 8, c->st));
    HIPCHK(c, c->fq_odd.ensure(std::max<uint64_t>(n, 1) * 8, false, 0, c->st));
    uint64_t *d_qs = nullptr, *d_qe = nullptr;
    if (tp) {
        HIPCHK(c, c->fq_qstart.ensure(std::max<uint64_t>(n, 1) * 8, false, 0, c->st));
        HIPCHK(c, c->fq_qend.ensure(std::max<uint64_t>(n, 1) * 8, false, 0, c->st));
        HIPCHK(c, hipMemsetAsync(c->fq_qstart.p, 0xFF, std::max<uint64_t>(n, 1) * 8, c->st));
        HIPCHK(c, hipMemsetAsync(c->fq_qend.p, 0xFF, std::max<uint64_t>(n, 1) * 8, c->st));
        d_qs = c->fq_qstart.as<uint64_t>(); d_qe = c->fq_qend.as<uint64_t>();
    }
    HIPCHK(c, hipMemsetAsync(c->ctr.p, 0, 128, c->st));
    // pass 2 per window: record starts.  Single-window files reuse the scan of pass 1.
    kt_begin(c, "fastq_index");
    uint64_t o0 = 0; size_t wi = 0;
    for (uint64_t w0 = 0; w0 < size; w0 += FQ_WINDOW, ++wi) {
        uint64_t wsz = std::min(FQ_WINDOW, size - w0);
        uint32_t nseg = (uint32_t)((wsz + FQ_SEG - 1) / FQ_SEG);
        if (size > FQ_WINDOW) {
            HIPCHK(c, hipMemsetAsync(c->fq_seg.as<uint32_t>() + nseg, 0, 4, c->st));
            hipLaunchKernelGGL(k_fq_count, dim3(nseg), dim3(256), 0, c->st, d + w0, wsz, c->fq_seg.as<uint32_t>());
            int rc = scan_exclusive(c, c->fq_seg.as<uint32_t>(), c->fq_segb.as<uint32_t>(), (uint64_t)nseg + 1);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_fq_index, dim3(nseg), dim3(256), 0, c->st, d + w0, wsz, w0, o0, c->fq_segb.as<uint32_t>(), c->fq_start.as<uint64_t>(), d_qs, d_qe);
        o0 += win_nl[wi];
    }
    kt_end(c, "fastq_index", size + n * 8);
    HIPCHK(c, hipGetLastError());
    if (n && tp) {
        // phred base as trim_file.py decides it: the workers use what record 0 says, miRge.pl is told what 1000 records say
        hipLaunchKernelGGL(k_fq_phred, dim3(4), dim3(256), 0, c->st, d, size, d_qs, d_qe, n, c->ctr.as<unsigned long long>() + 5);
        HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->ctr.p, 64, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        tp->phred_workers = c->h_ctr[5] ? 64 : 33; tp->phred_reported = c->h_ctr[6] ? 64 : 33;
        tp->dev.base = tp->phred_workers;
        kt_begin(c, "fastq_trim");
        const uint32_t tb = tp->dev.max_alen > 32 ? TRIM_BLOCK / 2 : TRIM_BLOCK;
        hipLaunchKernelGGL(k_fq_trim_pack, dim3((uint32_t)((n + tb - 1) / tb)), dim3(tb), (size_t)(tp->dev.max_alen + 1) * tb * 4, c->st, d, size,
                           c->fq_start.as<uint64_t>(), d_qs, d_qe, n, tp->dev, c->stage16.as<uint64_t>(), c->stage48.as<uint64_t>(),
                           c->fq_odd.as<uint64_t>(), c->ctr.as<unsigned long long>());
        kt_end(c, "fastq_trim", size);
        HIPCHK(c, hipGetLastError());
    } else if (n) {
        kt_begin(c, "fastq_pack");
        hipLaunchKernelGGL(k_fq_pack, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->st, d, size, c->fq_start.as<uint64_t>(), n,
                           c->stage16.as<uint64_t>(), c->stage48.as<uint64_t>(), c->fq_odd.as<uint64_t>(), c->ctr.as<unsigned long long>());
        kt_end(c, "fastq_pack", n * 24 + size / 4);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->ctr.p, 72, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (tp) tp->kept = c->h_ctr[4];
    if (c->h_ctr[2] == 6) return fail(c, MIRGE_E_FORMAT, "record " + std::to_string(c->h_ctr[3]) + ": sequence and quality lines differ in length");
    if (c->h_ctr[2] == 4 || c->h_ctr[2] == 5) return verb_refused(c, (int)c->h_ctr[2], c->h_ctr[3]);
    if (const uint64_t m = c->h_ctr[7]) {
        // reads no key holds: their bytes come back in one pool and join the verbatim table
        const uint64_t nbytes = c->h_ctr[8];
        HIPCHK(c, c->fq_oddtab.ensure(m * 8 + 8, false, 0, c->st));
        HIPCHK(c, c->fq_oddpool.ensure(nbytes + 8, false, 0, c->st));
        unsigned long long* cursor = c->fq_oddtab.as<unsigned long long>() + m;
        HIPCHK(c, hipMemsetAsync(cursor, 0, 8, c->st));
        hipLaunchKernelGGL(k_fq_odd_gather, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, c->st, d, c->fq_start.as<uint64_t>(),
                           c->fq_odd.as<uint64_t>(), m, c->fq_oddpool.as<uint8_t>(), c->fq_oddtab.as<uint64_t>(), cursor);
        HIPCHK(c, hipGetLastError());
        // table + bytes into ONE pinned buffer the context keeps (pageable vectors: 4 GB/s and a zero fill; pinned: the link's rate)
        const size_t need = (size_t)m * 8 + (size_t)nbytes + 16;
        if (c->h_odd_cap < need) {
            if (c->h_odd) (void)hipHostFree(c->h_odd);
            c->h_odd = nullptr; c->h_odd_cap = 0;
            HIPCHK(c, hipHostMalloc((void**)&c->h_odd, need + need / 4));
            c->h_odd_cap = need + need / 4;
        }
        const uint64_t* tab = reinterpret_cast<const uint64_t*>(c->h_odd);
        const char* pool = reinterpret_cast<const char*>(c->h_odd) + (size_t)m * 8;
        HIPCHK(c, hipMemcpyAsync(c->h_odd, c->fq_oddtab.p, m * 8, hipMemcpyDeviceToHost, c->st));
        if (nbytes) HIPCHK(c, hipMemcpyAsync(c->h_odd + (size_t)m * 8, c->fq_oddpool.p, nbytes, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        c->vb.add_batch(m, [&](uint64_t j) { return std::make_pair(pool + (tab[(size_t)j] >> 16), (size_t)(tab[(size_t)j] & 0xFFFF)); },
                        sample, c->S, host_threads());
    }
    uint64_t n16 = c->h_ctr[0], n48 = c->h_ctr[1];
    return collapse_packed(c, c->stage16.p, n16, c->stage48.p, n48, sample);
}

// gzip file -> HBM (the reference takes fastq or fastq.gz, miRge.pl:1092; cutadapt's reader inflates it inside
// trim_file.py:123).  A gzip stream inflates serially: one thread fills the pinned ring slot by slot while the DMA of
// the previous slots runs; the device buffer starts at the size the gzip trailer states and grows if the file holds more.
int fastq_gz_to_hbm(mirge_ctx* c, const char* path, uint64_t csize, uint64_t* size_out, uint8_t* last_out) {
    auto t0 = std::chrono::steady_clock::now();
    uint64_t hint = csize * 4;
    if (FILE* f = fopen(path, "rb")) {               // ISIZE: inflated size mod 2^32 of the last member
        unsigned char t[4];
        if (csize >= 18 && fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4) {
            uint64_t isz = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
            if (isz > hint) hint = isz;
        }
        fclose(f);
    }
    gzFile g = gzopen(path, "rb");
    if (!g) return fail(c, MIRGE_E_IO, std::string("cannot open ") + path);
    (void)gzbuffer(g, 1u << 20);
    if (!c->fq_pin && hipHostMalloc((void**)&c->fq_pin, FQ_SLOTS * FQ_SLOT_BYTES) != hipSuccess) {
        gzclose(g); return fail(c, MIRGE_E_HIP, "cannot allocate the pinned staging ring");
    }
    hipError_t he = c->fq_raw.ensure(hint + 256, false, 0, c->st);
    hipEvent_t ev[FQ_SLOTS];
    for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int rc = he == hipSuccess ? MIRGE_OK : fail(c, MIRGE_E_HIP, std::string("FASTQ buffer: ") + hipGetErrorString(he));
    uint64_t off = 0;
    *last_out = '\n';
    for (uint64_t i = 0; rc == MIRGE_OK; ++i) {
        const int slot = (int)(i % FQ_SLOTS);
        if (i >= (uint64_t)FQ_SLOTS && hipEventSynchronize(ev[slot]) != hipSuccess) { rc = fail(c, MIRGE_E_HIP, "FASTQ upload failed"); break; }
        uint8_t* dst = c->fq_pin + (size_t)slot * FQ_SLOT_BYTES;
        size_t got = 0;
        while (got < FQ_SLOT_BYTES) {
            int r = gzread(g, dst + got, (unsigned)std::min<size_t>(FQ_SLOT_BYTES - got, 1u << 30));
            if (r < 0) { int en = 0; const char* m = gzerror(g, &en); rc = fail(c, MIRGE_E_IO, std::string(path) + ": " + (m ? m : "gzip error")); break; }
            if (r == 0) break;
            got += (size_t)r;
        }
        if (rc || got == 0) break;
        he = c->fq_raw.ensure(off + got + 256, true, off, c->st);       // grows by half when the trailer understated the size
        if (he == hipSuccess) he = hipMemcpyAsync(c->fq_raw.as<uint8_t>() + off, dst, got, hipMemcpyHostToDevice, c->st);
        if (he == hipSuccess) he = hipEventRecord(ev[slot], c->st);
        if (he != hipSuccess) { rc = fail(c, MIRGE_E_HIP, std::string("FASTQ upload: ") + hipGetErrorString(he)); break; }
        *last_out = dst[got - 1];
        off += got;
        if (got < FQ_SLOT_BYTES) break;
    }
    if (rc == MIRGE_OK) {
        int en = 0; (void)gzerror(g, &en);
        if (en != Z_OK && en != Z_STREAM_END) rc = fail(c, MIRGE_E_IO, std::string(path) + ": corrupt or truncated gzip stream");
    }
    gzclose(g);
    hipError_t e = hipStreamSynchronize(c->st);
    for (auto& x : ev) (void)hipEventDestroy(x);
    if (rc == MIRGE_OK && e != hipSuccess) rc = fail(c, MIRGE_E_HIP, std::string("FASTQ upload: ") + hipGetErrorString(e));
    *size_out = off;
    c->ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// file -> HBM: reader threads pread() slices into a pinned ring, this thread queues one DMA per slice in file order
int fastq_file_to_hbm(mirge_ctx* c, const char* path, uint64_t* size_out, uint8_t* last_out) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(c, MIRGE_E_IO, std::string("cannot open ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { close(fd); return fail(c, MIRGE_E_IO, std::string(path) + " is not a regular file"); }
    const uint64_t size = (uint64_t)sb.st_size;
    {   // gzip magic?
        unsigned char mg[2] = {0, 0};
        if (size >= 2 && pread(fd, mg, 2, 0) == 2 && mg[0] == 0x1f && mg[1] == 0x8b) { close(fd); return fastq_gz_to_hbm(c, path, size, size_out, last_out); }
    }
    *size_out = size; *last_out = '\n';
    auto t0 = std::chrono::steady_clock::now();
    hipError_t he = c->fq_raw.ensure(size + 256, false, 0, c->st);
    if (he != hipSuccess) { close(fd); return fail(c, MIRGE_E_HIP, std::string("FASTQ buffer: ") + hipGetErrorString(he)); }
    if (size == 0) { close(fd); return MIRGE_OK; }
    if (!c->fq_pin && hipHostMalloc((void**)&c->fq_pin, FQ_SLOTS * FQ_SLOT_BYTES) != hipSuccess) {
        close(fd); return fail(c, MIRGE_E_HIP, "cannot allocate the pinned staging ring");
    }
    const uint64_t nchunk = (size + FQ_SLOT_BYTES - 1) / FQ_SLOT_BYTES;
    std::mutex mu; std::condition_variable cv;
    std::vector<char> filled((size_t)nchunk, 0);
    uint64_t freed = 0;                              // chunks whose DMA has completed (their slots may be refilled)
    std::atomic<uint64_t> next{0};
    bool io_error = false, stop = false;
    auto reader = [&]() {
        for (;;) {
            uint64_t i = next.fetch_add(1);
            if (i >= nchunk) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || i < freed + FQ_SLOTS; }); if (stop) return; }
            uint8_t* dst = c->fq_pin + (i % FQ_SLOTS) * FQ_SLOT_BYTES;
            uint64_t off = i * FQ_SLOT_BYTES, want = std::min<uint64_t>(FQ_SLOT_BYTES, size - off), got = 0;
            while (got < want) {
                ssize_t r = pread(fd, dst + got, (size_t)(want - got), (off_t)(off + got));
                if (r <= 0) break;
                got += (uint64_t)r;
            }
            { std::lock_guard<std::mutex> lk(mu); if (got != want) io_error = true; filled[(size_t)i] = 1; }
            cv.notify_all();
        }
    };
    const int nthr = (int)std::min<uint64_t>(4, nchunk);
    std::vector<std::thread> thr;
    for (int t = 0; t < nthr; ++t) thr.emplace_back(reader);
    hipEvent_t ev[FQ_SLOTS];
    for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int rc = MIRGE_OK;
    for (uint64_t i = 0; i < nchunk && rc == MIRGE_OK; ++i) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[(size_t)i] != 0; }); if (io_error) rc = fail(c, MIRGE_E_IO, std::string("short read on ") + path); }
        if (rc) break;
        uint64_t off = i * FQ_SLOT_BYTES, len = std::min<uint64_t>(FQ_SLOT_BYTES, size - off);
        const uint8_t* src = c->fq_pin + (i % FQ_SLOTS) * FQ_SLOT_BYTES;
        if (i + 1 == nchunk) *last_out = src[len - 1];
        hipError_t e = hipMemcpyAsync(c->fq_raw.as<uint8_t>() + off, src, len, hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipEventRecord(ev[i % FQ_SLOTS], c->st);
        if (e == hipSuccess && i >= 1) e = hipEventSynchronize(ev[(i - 1) % FQ_SLOTS]);      // chunk i is queued behind it: the DMA engine stays busy
        if (e != hipSuccess) { rc = fail(c, MIRGE_E_HIP, std::string("FASTQ upload: ") + hipGetErrorString(e)); break; }
        if (i >= 1) { { std::lock_guard<std::mutex> lk(mu); freed = i; } cv.notify_all(); }
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : thr) t.join();
    hipError_t e = hipStreamSynchronize(c->st);
    for (auto& x : ev) (void)hipEventDestroy(x);
    close(fd);
    if (rc == MIRGE_OK && e != hipSuccess) rc = fail(c, MIRGE_E_HIP, std::string("FASTQ upload: ") + hipGetErrorString(e));
    c->ingest_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}


// ---- multi-GPU exchange step (shard.h): export this context's (unique, sample, count) entries grouped by owner
// rank, import the records a rank receives.  The all-to-all itself is the caller's (torch.distributed / RCCL).
template <int KW>
void shard_count_class(mirge_ctx* c, Class& C, int world, unsigned long long* d_cnt) {
    for (auto& sg : C.segs) {
        if (!sg.count) continue;
        hipLaunchKernelGGL((k_shard_count<KW>), dim3((uint32_t)((sg.count + 255) / 256)), dim3(256), 0, c->st,
                           C.coo_g.as<uint32_t>() + sg.start, (uint32_t)sg.count, C.ukeys.as<uint64_t>(), (uint32_t)world, d_cnt);
    }
}
template <int KW>
void shard_scatter_class(mirge_ctx* c, Class& C, int world, unsigned long long* d_cur, void* d_rec) {
    for (auto& sg : C.segs) {
        if (!sg.count) continue;
        hipLaunchKernelGGL((k_shard_scatter<KW>), dim3((uint32_t)((sg.count + 255) / 256)), dim3(256), 0, c->st,
                           C.coo_g.as<uint32_t>() + sg.start, C.coo_c.as<uint32_t>() + sg.start, (uint32_t)sg.count, (uint32_t)sg.sample,
                           C.ukeys.as<uint64_t>(), (uint32_t)world, d_cur, (uint64_t*)d_rec);
    }
}

int shard_export_sizes(mirge_ctx* c, int world, uint64_t* n16, uint64_t* n48) {
    if (world < 1 || world > SHARD_MAX_WORLD) return fail(c, MIRGE_E_ARG, "world size must be between 1 and " + std::to_string(SHARD_MAX_WORLD));
    HIPCHK(c, c->sh_ctr.ensure(2 * SHARD_MAX_WORLD * 8, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->sh_ctr.p, 0, 2 * SHARD_MAX_WORLD * 8, c->st));
    shard_count_class<2>(c, c->cls[0], world, c->sh_ctr.as<unsigned long long>());
    shard_count_class<6>(c, c->cls[1], world, c->sh_ctr.as<unsigned long long>() + SHARD_MAX_WORLD);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&c->sh_n[0][0], c->sh_ctr.p, 2 * SHARD_MAX_WORLD * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    c->sh_world = world;
    for (int d = 0; d < world; ++d) { if (n16) n16[d] = c->sh_n[0][d]; if (n48) n48[d] = c->sh_n[1][d]; }
    return MIRGE_OK;
}

int shard_export(mirge_ctx* c, int world, void* d16, void* d48) {
    if (world != c->sh_world) return fail(c, MIRGE_E_ARG, "mirge_shard_export: call mirge_shard_export_sizes with the same world size first");
    uint64_t cur[2][SHARD_MAX_WORLD] = {};
    uint64_t tot[2] = {0, 0};
    for (int k = 0; k < 2; ++k) for (int d = 0; d < world; ++d) { cur[k][d] = tot[k]; tot[k] += c->sh_n[k][d]; }
    if ((tot[0] && !d16) || (tot[1] && !d48)) return fail(c, MIRGE_E_ARG, "null device buffer");
    HIPCHK(c, hipMemcpyAsync(c->sh_ctr.p, &cur[0][0], sizeof cur, hipMemcpyHostToDevice, c->st));
    shard_scatter_class<2>(c, c->cls[0], world, c->sh_ctr.as<unsigned long long>(), d16);
    shard_scatter_class<6>(c, c->cls[1], world, c->sh_ctr.as<unsigned long long>() + SHARD_MAX_WORLD, d48);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->st));       // cur[] lives on this stack frame
    return MIRGE_OK;
}

// records of one class -> keys + weights grouped by sample (in `keys` / `wt`), per-sample counts in cnt_out
template <int KW>
int group_records(mirge_ctx* c, const void* d_rec, uint64_t n, DevBuf& keys, DevBuf& wt, std::vector<uint64_t>& cnt_out) {
    const int S = c->S;
    cnt_out.assign((size_t)S, 0);
    if (n == 0) return MIRGE_OK;
    if (n >= 0xFFFFFFF0ull) return fail(c, MIRGE_E_LIMIT, "more than 2^32 records in one import");
    HIPCHK(c, c->sh_ctr.ensure(std::max<size_t>((size_t)(S + 1) * 8, 2 * SHARD_MAX_WORLD * 8), false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->sh_ctr.p, 0, (size_t)(S + 1) * 8, c->st));
    HIPCHK(c, keys.ensure(n * KW * 8, false, 0, c->st));
    HIPCHK(c, wt.ensure(n * 4, false, 0, c->st));
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL((k_rec_count<KW>), dim3(grid), dim3(256), 0, c->st, (const uint64_t*)d_rec, n, (uint32_t)S,
                       c->sh_ctr.as<unsigned long long>(), c->sh_ctr.as<unsigned long long>() + S);
    HIPCHK(c, hipGetLastError());
    std::vector<uint64_t> h((size_t)S + 1);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->sh_ctr.p, (size_t)(S + 1) * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (h[(size_t)S]) return fail(c, MIRGE_E_ARG, std::to_string(h[(size_t)S]) + " imported records name a sample outside this context");
    std::vector<uint64_t> cur((size_t)S);
    uint64_t run = 0;
    for (int s = 0; s < S; ++s) { cnt_out[(size_t)s] = h[(size_t)s]; cur[(size_t)s] = run; run += h[(size_t)s]; }
    HIPCHK(c, hipMemcpyAsync(c->sh_ctr.p, cur.data(), (size_t)S * 8, hipMemcpyHostToDevice, c->st));
    hipLaunchKernelGGL((k_rec_scatter<KW>), dim3(grid), dim3(256), 0, c->st, (const uint64_t*)d_rec, n, (uint32_t)S,
                       c->sh_ctr.as<unsigned long long>(), keys.as<uint64_t>(), wt.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->st));       // cur lives on this stack frame
    return MIRGE_OK;
}

int shard_import(mirge_ctx* c, const void* d16, uint64_t n16, const void* d48, uint64_t n48) {
    if (c->S > REC_LDS_S) return fail(c, MIRGE_E_LIMIT, "more than " + std::to_string(REC_LDS_S) + " samples in a sharded run");
    if ((n16 && !d16) || (n48 && !d48)) return fail(c, MIRGE_E_ARG, "null device buffer");
    std::vector<uint64_t> c16, c48;
    int rc = group_records<2>(c, d16, n16, c->stage16, c->sh_w[0], c16);
    if (!rc) rc = group_records<6>(c, d48, n48, c->stage48, c->sh_w[1], c48);
    if (rc) return rc;
    for (auto& C : c->cls) C.reset();
    c->annotated = false; c->counts_valid = false; c->host_valid = false;
    kt_clear(c, "collapse");
    uint64_t o16 = 0, o48 = 0;
    for (int s = 0; s < c->S && !rc; ++s) {
        rc = collapse_both(c, c->stage16.as<uint64_t>() + o16 * 2, c16[(size_t)s], c->sh_w[0].as<uint32_t>() + o16,
                           c->stage48.as<uint64_t>() + o48 * 6, c48[(size_t)s], c->sh_w[1].as<uint32_t>() + o48, s);
        o16 += c16[(size_t)s]; o48 += c48[(size_t)s];
    }
    if (!rc) { HIPCHK(c, hipStreamSynchronize(c->st)); kt_resolve(c); }
    return rc;
}

// ---- cascade + count vectors
// do rounds 1 and 2 each walk their own length class (Class::ll) ?
bool ll_rounds(const mirge_ctx* c, const Class& C) {
    if (C.KW != 2 || !C.ll_ok || !C.ll_fresh || c->n_rounds < 2) return false;
    if (const char* e = getenv("MIRGE_LEN_LISTS")) { if (atoi(e) == 0) return false; }       // A/B aid
    const mirge_round &a = c->rounds[0], &b = c->rounds[1];
    return a.len_lt == LL_SPLIT && a.len_gt == 0 && b.len_gt == LL_SPLIT - 1 && b.len_lt == 0;
}
// the table's size as the launches see it: with a deferred collapse behind them an upper bound, the real one on the device
inline uint64_t table_bound(const Class& C) { return C.U + ((C.pending && C.pend_n) ? C.pend_n : 0); }
inline const unsigned long long* table_dev(const Class& C) { return (C.pending && C.pend_n) ? C.ws_ctr.as<unsigned long long>() : (const unsigned long long*)nullptr; }
template <int KW, int RN>
void launch_match(mirge_ctx* c, Class& C, const RoundDev& P, const LibDev& L, int thr, int r) {
    // round 1 keeps no survivor list (nearly everything survives it), round 2 scans the table and writes the first one;
    // with length-class lists rounds 1 and 2 walk one each (full lanes, and both meet their uniques for the first
    // time: init), and round 3 is the one that scans the table -- 93 % of it is still unannotated by then
    const bool ll = ll_rounds(c, C);
    const uint64_t UB = table_bound(C);
    const unsigned long long* U_dev = table_dev(C);
    unsigned long long* const cnt_in = C.act_n + (size_t)r * LIST_K * LIST_NSTRIDE;
    unsigned long long* const cnt_out = C.act_n + (size_t)(r + 1) * LIST_K * LIST_NSTRIDE;
    ListIn in{nullptr, nullptr, 0, 0};
    if (r <= 1) { if (ll) in = ListIn{C.ll[r].as<uint32_t>(), C.ll_ctr.as<unsigned long long>() + r * LL_CSTRIDE, 1, 0}; }
    else if (!(r == 2 && ll)) in = ListIn{C.act[(r + 1) & 1].as<uint32_t>(), cnt_in, C.lf[r].K, C.lf[r].cap};
    const bool writes = (ll ? r >= 2 : r >= 1) && r + 1 < c->n_rounds;
    const bool init = r == 0 || (r == 1 && ll);
    // the lists of the generic kernel's own leftovers (16-byte class, probe rounds): read whenever an earlier round wrote any
    const bool had_list = ll ? r >= 3 : r >= 2;
    const uint32_t* fbs_in = had_list ? C.fbs[(r + 1) & 1].as<uint32_t>() : (const uint32_t*)nullptr;
    if (KW == 2 && lds_round(L, P)) {
        // the LDS path (kernels.h: k_lds_round): persistent blocks with the library's text in LDS, then the generic
        // kernel over the reads it handed on
        const RoundSch RS = round_schemes(L, P);
        auto kern = RS.n == 0 ? k_lds_round<RN, 0> : ((RS.n == 1 && blocks_equal(tab_blocks(RS.t[0]))) ? k_lds_round<RN, 1> : k_lds_round<RN, ROUND_SCHEMES>);      // (<.., 1>: one scheme, of equal blocks)
        const int LDS_BLOCK = RS.n == 0 ? LdsCfg<0>::BLOCK : LdsCfg<1>::BLOCK, LDS_WAVES = LDS_BLOCK / 64;
        const size_t lds = lds_lib_bytes(L) + (size_t)LDS_WAVES * sizeof(LdsWave);
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const uint64_t tiles = in.K > 1 ? (uint64_t)in.K * (in.cap / 64) : (UB + 63) / 64;
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160u << 10) / (lds + 512)));
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->n_cu * per_cu, (tiles + LDS_WAVES - 1) / LDS_WAVES));
        const uint32_t cap_out = (uint32_t)(((tiles + LIST_K - 1) / LIST_K * 64 + PB - 1) / PB * PB);      // tiles that share a sub-list x 64, in whole blocks of the reader
        ListOut out{nullptr, nullptr, 0, 0};
        if (writes) { out = ListOut{C.act[r & 1].as<uint32_t>(), cnt_out, (uint32_t)LIST_K, cap_out}; C.lf[r + 1].K = LIST_K; C.lf[r + 1].cap = cap_out; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(LDS_BLOCK), lds, C.st,
                           C.ukeys.as<uint64_t>(), (uint32_t)UB, C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(), C.ann_amb.as<uint8_t>(),
                           L, P, RS, c->d_stats + (size_t)r * STAT_WORDS, in, out,
                           C.fb.as<uint32_t>(), C.fb_n + r, init, U_dev, (uint32_t)C.U);
        hipLaunchKernelGGL((k_match_round<KW, RN>), dim3((uint32_t)std::min<uint64_t>((UB + MB - 1) / MB, 512)), dim3(MB), 0, C.st,
                           C.ukeys.as<uint64_t>(), (uint32_t)UB, C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(),
                           C.ann_amb.as<uint8_t>(), L, P, thr, c->d_stats + (size_t)r * STAT_WORDS,
                           ListIn{C.fb.as<uint32_t>(), C.fb_n + r, 1, 0}, fbs_in, (const unsigned long long*)(C.fbs_n + r),
                           writes ? C.fbs[r & 1].as<uint32_t>() : (uint32_t*)nullptr, C.fbs_n + r + 1, false,
                           U_dev, (uint32_t)C.U);
        return;
    }
    if (probe_round(L, P)) {
        // the probe path (kernels.h: k_probe_round) for what the slot tables can serve, then the generic kernel over the
        // reads it handed on.  The 48-byte class (a few ten thousand uniques) takes one instantiation for any number of schemes.
        const RoundSch RS = round_schemes(L, P);
        auto kern2 = RS.n == 0 ? k_probe_round<RN, 0> : ((RS.n == 1 && blocks_equal(tab_blocks(RS.t[0]))) ? k_probe_round<RN, 1> : k_probe_round<RN, ROUND_SCHEMES>);      // (<.., 1>: one scheme, of equal blocks)
        auto kern = KW == 2 ? kern2 : k_probe_round<RN, ROUND_SCHEMES, 6>;
        const uint32_t grid = in.K > 1 ? in.K * (in.cap / PB) : (uint32_t)((UB + PB - 1) / PB);
        const uint32_t cap_out = (grid + LIST_K - 1) / LIST_K * PB;          // blocks that share a sub-list x what each can append
        ListOut out{nullptr, nullptr, 0, 0};
        if (writes) { out = ListOut{C.act[r & 1].as<uint32_t>(), cnt_out, (uint32_t)LIST_K, cap_out}; C.lf[r + 1].K = LIST_K; C.lf[r + 1].cap = cap_out; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PB), 0, C.st,
                           C.ukeys.as<uint64_t>(), (uint32_t)UB, C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(), C.ann_amb.as<uint8_t>(),
                           L, P, RS, c->d_stats + (size_t)r * STAT_WORDS, in, out,
                           C.fb.as<uint32_t>(), C.fb_n + r, init, U_dev, (uint32_t)C.U);
        // (a small grid: its blocks stride over the lists, whose lengths -- usually a handful -- only the device knows)
        hipLaunchKernelGGL((k_match_round<KW, RN>), dim3((uint32_t)std::min<uint64_t>((UB + MatchTile<KW>::N - 1) / MatchTile<KW>::N, 512)), dim3(MB), 0, C.st,
                           C.ukeys.as<uint64_t>(), (uint32_t)UB, C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(),
                           C.ann_amb.as<uint8_t>(), L, P, thr, c->d_stats + (size_t)r * STAT_WORDS,
                           ListIn{C.fb.as<uint32_t>(), C.fb_n + r, 1, 0}, fbs_in, (const unsigned long long*)(C.fbs_n + r),
                           writes ? C.fbs[r & 1].as<uint32_t>() : (uint32_t*)nullptr, C.fbs_n + r + 1, false,
                           U_dev, (uint32_t)C.U);
        return;
    }
    // the generic kernel as the round's matcher: it reads whatever layout the previous round left (and the leftovers'
    // list), and writes one dense list
    if (writes) { C.lf[r + 1].K = 1; C.lf[r + 1].cap = 0; }
    hipLaunchKernelGGL((k_match_round<KW, RN>), dim3((uint32_t)((UB + MatchTile<KW>::N - 1) / MatchTile<KW>::N)), dim3(MB), 0, C.st,
                       C.ukeys.as<uint64_t>(), (uint32_t)UB, C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(),
                       C.ann_amb.as<uint8_t>(), L, P, thr, c->d_stats + (size_t)r * STAT_WORDS,
                       in, fbs_in, (const unsigned long long*)(C.fbs_n + r),
                       writes ? C.act[r & 1].as<uint32_t>() : (uint32_t*)nullptr, cnt_out, init, U_dev, (uint32_t)C.U);
}

// the round number is a template parameter only so that every round is its own kernel symbol in a
// rocprof trace (the library's own per-round timing and the trace can then be compared)
template <int KW>
int annotate_class(mirge_ctx* c, Class& C, int r, const mirge_round& R) {
    if (table_bound(C) == 0) return MIRGE_OK;
    RoundDev P{R.lib, R.len_lt, R.len_gt, R.trim5, R.trim3, R.seedlen, R.seedmm, R.maxmm};
    LibDev L = c->libs[R.lib].dev();
    int thr = round_np(P) >= 3 ? c->coop_thr3 : c->coop_thr;
    switch (r) {
        case 0: launch_match<KW, 0>(c, C, P, L, thr, r); break;
        case 1: launch_match<KW, 1>(c, C, P, L, thr, r); break;
        case 2: launch_match<KW, 2>(c, C, P, L, thr, r); break;
        case 3: launch_match<KW, 3>(c, C, P, L, thr, r); break;
        case 4: launch_match<KW, 4>(c, C, P, L, thr, r); break;
        case 5: launch_match<KW, 5>(c, C, P, L, thr, r); break;
        case 6: launch_match<KW, 6>(c, C, P, L, thr, r); break;
        default: launch_match<KW, 7>(c, C, P, L, thr, r); break;
    }
    HIPCHK(c, hipGetLastError());
    return MIRGE_OK;
}

// fewer entries than this: global atomics, no LDS columns (kernels.h: k_counts).  MIRGE_CNT_LDS_MIN: tests put the LDS form on small inputs
inline uint32_t counts_block() { if (const char* e = getenv("MIRGE_CNT_BLOCK")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) return (uint32_t)v; } return 1024; }      // (A/B aid)
inline uint64_t cnt_lds_min() { if (const char* e = getenv("MIRGE_CNT_LDS_MIN")) return strtoull(e, nullptr, 10); return 1u << 18; }
template <int KW>
int counts_class(mirge_ctx* c, Class& C, int M) {
    const uint64_t CNT_LDS_MIN_ENTRIES = cnt_lds_min();
    uint32_t round_libs = 0;
    for (int r = 0; r < c->n_rounds; ++r) round_libs |= (uint32_t)(c->rounds[r].lib & 3) << (2 * r);
    const uint32_t iscan_rounds = c->rounds[0].lib == MIRGE_LIB_MIRNA ? 1u : 0u;      // miRge.pl:472-475: annot 1 only
    for (auto& sg : C.segs) {
        if (!sg.count) continue;
        const size_t lds = (M <= CNT_LDS_M && sg.count >= CNT_LDS_MIN_ENTRIES) ? (size_t)2 * M * 8 : 0;
        // with the columns in LDS (42 KB at a human-sized library) three blocks fit a CU: blocks of 1024 threads then, so
        // that the CU still runs 2048 of them (blocks of 256: 3 waves per SIMD, 0.118 ms; 512: 0.09; 1024: 0.072)
        const uint32_t cb = lds ? counts_block() : 256u;
        const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / cb, (150u << 10) / (lds + 2048)));      // blocks a CU holds
        uint32_t grid = (uint32_t)std::min<uint64_t>(256 * per_cu, (sg.count + cb - 1) / cb);
        hipLaunchKernelGGL((k_counts<KW>), dim3(grid), dim3(cb), lds, C.st, C.coo_g.as<uint32_t>() + sg.start,
                           C.coo_c.as<uint32_t>() + sg.start, (uint32_t)sg.count, sg.sample, c->S, C.ukeys.as<uint64_t>(),
                           C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(), M, round_libs, iscan_rounds,
                           c->d_quant, c->d_iscan, c->d_cat, c->d_len, c->d_tu, (const unsigned long long*)nullptr, lds != 0);
    }
    if (C.pending && C.pend_n) {
        // the sample of a deferred collapse: its entries follow the settled ones, how many there are the kernel reads itself
        const size_t lds = (M <= CNT_LDS_M && C.pend_n >= CNT_LDS_MIN_ENTRIES) ? (size_t)2 * M * 8 : 0;
        const uint32_t cb = lds ? counts_block() : 256u;
        const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(2048 / cb, (150u << 10) / (lds + 2048)));
        uint32_t grid = (uint32_t)std::min<uint64_t>(256 * per_cu, ((uint64_t)C.pend_n + cb - 1) / cb);
        hipLaunchKernelGGL((k_counts<KW>), dim3(grid), dim3(cb), lds, C.st, C.coo_g.as<uint32_t>() + C.coo_n,
                           C.coo_c.as<uint32_t>() + C.coo_n, C.pend_n, C.pend_sample, c->S, C.ukeys.as<uint64_t>(),
                           C.ann_round.as<uint8_t>(), C.ann_ref.as<int32_t>(), M, round_libs, iscan_rounds,
                           c->d_quant, c->d_iscan, c->d_cat, c->d_len, c->d_tu, C.ws_ctr.as<unsigned long long>(), lds != 0);
    }
    HIPCHK(c, hipGetLastError());
    return MIRGE_OK;
}

// the verbatim uniques: pool of packed records up, every round, their share of the count vectors (on st2, behind the
// 48-byte class)
int annotate_verb(mirge_ctx* c) {
    Verb& V = c->vb;
    const uint64_t U = V.seq.size(), E = V.coo_u.size();
    if (U == 0) return MIRGE_OK;
    hipStream_t st = c->st2;
    std::vector<uint64_t> off((size_t)U); std::vector<uint32_t> len((size_t)U);
    uint64_t words = 0;
    for (uint64_t u = 0; u < U; ++u) { off[(size_t)u] = words; len[(size_t)u] = (uint32_t)V.seq[(size_t)u].size(); words += (uint64_t)verb_words((int)len[(size_t)u]); }
    std::vector<uint64_t> pool((size_t)words);
    parallel_slices(U, host_threads(), [&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t u = lo; u < hi; ++u) verb_pack(V.seq[(size_t)u].data(), (int)len[(size_t)u], pool.data() + off[(size_t)u]);
    });
    HIPCHK(c, V.pool.ensure(words * 8, false, 0, st)); HIPCHK(c, V.off.ensure(U * 8, false, 0, st)); HIPCHK(c, V.len.ensure(U * 4, false, 0, st));
    HIPCHK(c, V.ann_round.ensure(U, false, 0, st)); HIPCHK(c, V.ann_ref.ensure(U * 4, false, 0, st)); HIPCHK(c, V.ann_amb.ensure(U, false, 0, st));
    HIPCHK(c, V.d_u.ensure(E * 4 + 4, false, 0, st)); HIPCHK(c, V.d_s.ensure(E * 4 + 4, false, 0, st)); HIPCHK(c, V.d_c.ensure(E * 4 + 4, false, 0, st));
    HIPCHK(c, hipMemcpyAsync(V.pool.p, pool.data(), words * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(V.off.p, off.data(), U * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(V.len.p, len.data(), U * 4, hipMemcpyHostToDevice, st));
    if (E) {
        HIPCHK(c, hipMemcpyAsync(V.d_u.p, V.coo_u.data(), E * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(V.d_s.p, V.coo_s.data(), E * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(V.d_c.p, V.coo_c.data(), E * 4, hipMemcpyHostToDevice, st));
    }
    uint32_t round_libs = 0;
    for (int r = 0; r < c->n_rounds; ++r) round_libs |= (uint32_t)(c->rounds[r].lib & 3) << (2 * r);
    const uint32_t iscan_rounds = c->rounds[0].lib == MIRGE_LIB_MIRNA ? 1u : 0u;
    for (int r = 0; r < c->n_rounds; ++r) {
        const mirge_round& R = c->rounds[r];
        RoundDev P{R.lib, R.len_lt, R.len_gt, R.trim5, R.trim3, R.seedlen, R.seedmm, R.maxmm};
        hipLaunchKernelGGL(k_match_verb, dim3((uint32_t)((U + 255) / 256)), dim3(256), 0, st, V.pool.as<uint64_t>(), V.off.as<uint64_t>(),
                           V.len.as<uint32_t>(), (uint32_t)U, V.ann_round.as<uint8_t>(), V.ann_ref.as<int32_t>(), V.ann_amb.as<uint8_t>(),
                           c->libs[R.lib].dev(), P, r, c->d_stats + (size_t)r * STAT_WORDS);
    }
    if (E)
        hipLaunchKernelGGL(k_counts_verb, dim3((uint32_t)((E + 255) / 256)), dim3(256), 0, st, V.d_u.as<uint32_t>(), V.d_s.as<uint32_t>(),
                           V.d_c.as<uint32_t>(), (uint32_t)E, c->S, V.len.as<uint32_t>(), V.ann_round.as<uint8_t>(), V.ann_ref.as<int32_t>(),
                           round_libs, iscan_rounds, LEN_BINS, c->d_quant, c->d_iscan, c->d_cat, c->d_len, c->d_tu);
    HIPCHK(c, hipGetLastError());
    // pool / off / len are host vectors of this frame: the copies must have left them
    HIPCHK(c, hipStreamSynchronize(st));
    return MIRGE_OK;
}

static const char* kRoundNames48[MIRGE_MAX_ROUNDS] = {"match_round1_48", "match_round2_48", "match_round3_48", "match_round4_48",
                                                      "match_round5_48", "match_round6_48", "match_round7_48", "match_round8_48"};
static const char* kRoundNames[MIRGE_MAX_ROUNDS] = {"match_round1", "match_round2", "match_round3", "match_round4",
                                                    "match_round5", "match_round6", "match_round7", "match_round8"};

int annotate(mirge_ctx* c) {
    for (int r = 0; r < c->n_rounds; ++r) {
        int lib = c->rounds[r].lib;
        if (lib < 0 || lib >= MIRGE_N_LIBS || !c->libs[lib].loaded)
            return fail(c, MIRGE_E_ARG, "round " + std::to_string(r + 1) + " uses a library that is not loaded");
    }
    if (!c->libs[MIRGE_LIB_MIRNA].loaded) return fail(c, MIRGE_E_ARG, "mirna library not loaded");
    for (int r = 0; r < c->n_rounds; ++r) {
        const mirge_round& R = c->rounds[r];
        int rc = ensure_schemes(c, c->libs[R.lib], R); if (rc) return rc;
    }
    kt_clear(c, "match"); kt_clear(c, "counts");
    // Round 1 walks the whole table and leaves every unique's annotation state behind (k_match_round<KW, 0>), so
    // ann_round / ann_ref / ann_amb need no fill of their own.
    for (int k = 0; k < 2; ++k) {
        Class& C = c->cls[k];
        size_t U = std::max<uint64_t>(table_bound(C), 1);
        HIPCHK(c, C.ann_round.ensure(U, false, 0, c->st));
        HIPCHK(c, C.ann_ref.ensure(U * 4, false, 0, c->st));
        HIPCHK(c, C.ann_amb.ensure(U, false, 0, c->st));
        HIPCHK(c, C.act[0].ensure((U + (size_t)LIST_K * 2 * PB) * 4, false, 0, c->st));      // (a split list: LIST_K sub-lists, each rounded up to whole tiles, then to whole blocks)
        HIPCHK(c, C.act[1].ensure((U + (size_t)LIST_K * 2 * PB) * 4, false, 0, c->st));
        HIPCHK(c, C.fb.ensure(U * 4, false, 0, c->st));
        HIPCHK(c, C.fbs[0].ensure(U * 4, false, 0, c->st));
        HIPCHK(c, C.fbs[1].ensure(U * 4, false, 0, c->st));
        for (auto& f : C.lf) f = Class::ListFmt{};
    }
    int M = (int)c->libs[MIRGE_LIB_MIRNA].names.size(), S = c->S;
    size_t qn = std::max<size_t>((size_t)M * S, 1);
    const size_t w_stats = MIRGE_MAX_ROUNDS * STAT_WORDS, w_cat = (size_t)5 * S, w_len = (size_t)LEN_BINS * S, w_tu = (size_t)S;
    const size_t w_act = (size_t)(MIRGE_MAX_ROUNDS + 1) * LIST_K * LIST_NSTRIDE;        // list length words of one key class: LIST_K per round, 128 B apart
    const size_t w_fb = MIRGE_MAX_ROUNDS + (MIRGE_MAX_ROUNDS + 1);                        // per class: fb list lengths, leftover-list lengths
    const size_t w_res = w_stats + 2 * qn + w_cat + w_len + w_tu;                       // the results ...
    const size_t w_out = w_res + 2 * w_fb, w_all = w_out + 2 * w_act;                   // ... + the fb list lengths: what ONE copy brings to the host
    HIPCHK(c, c->state.ensure(w_all * 8, false, 0, c->st));
    if (c->h_state_cap < w_out) {
        if (c->h_state) (void)hipHostFree(c->h_state);
        c->h_state = nullptr; c->h_state_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->h_state, w_out * 8));
        c->h_state_cap = w_out;
    }
    c->d_stats = c->state.as<unsigned long long>();
    c->d_quant = c->d_stats + w_stats; c->d_iscan = c->d_quant + qn; c->d_cat = c->d_iscan + qn;
    c->d_len = c->d_cat + w_cat; c->d_tu = c->d_len + w_len;
    c->cls[0].fb_n = c->d_tu + w_tu; c->cls[0].fbs_n = c->cls[0].fb_n + MIRGE_MAX_ROUNDS;
    c->cls[1].fb_n = c->cls[0].fb_n + w_fb; c->cls[1].fbs_n = c->cls[1].fb_n + MIRGE_MAX_ROUNDS;
    c->cls[0].act_n = c->cls[1].fb_n + w_fb; c->cls[1].act_n = c->cls[0].act_n + w_act;
    zero_regions(c->st, c->state.as<unsigned long long>(), w_all);
    hipEvent_t e0 = c->ev_t[0], e1 = c->ev_t[1], e2 = c->ev_t[2];
    if (c->timing) HIPCHK(c, hipEventRecord(e0, c->st));
    // the 48-byte class (reads with N or longer than 56 nt: a fraction of a percent) runs its cascade on st2 beside
    // the 16-byte class; the two meet only in atomically updated counters.  Kernel times are those of the 16-byte class.
    { int rc = fork_st2(c); if (rc) return rc; }
    for (int r = 0; r < c->n_rounds; ++r) {
        kt_begin(c, kRoundNames[r]);
        int rc = annotate_class<2>(c, c->cls[0], r, c->rounds[r]);
        kt_end(c, kRoundNames[r], 0);
        // the 48-byte class's launches are timed by name as well (on their own stream): they overlap the 16-byte class's,
        // so their times are not summands of the step -- but whichever chain is longer is what the join waits for
        const bool t48 = table_bound(c->cls[1]) > 0;
        if (t48) kt_begin(c, kRoundNames48[r], c->st2);
        if (!rc) rc = annotate_class<6>(c, c->cls[1], r, c->rounds[r]);
        if (t48) kt_end(c, kRoundNames48[r], (uint64_t)52 * table_bound(c->cls[1]), c->st2);      // (bytes: an upper bound -- the round statistics do not tell the classes apart)
        if (rc) return rc;
    }
    if (c->timing) HIPCHK(c, hipEventRecord(e1, c->st));
    kt_begin(c, "counts");
    int rc = counts_class<2>(c, c->cls[0], M);
    kt_end(c, "counts", (c->cls[0].coo_n + c->cls[1].coo_n) * 8 + (uint64_t)16 * M * S);
    if (!rc && table_bound(c->cls[1]) > 0) {
        kt_begin(c, "counts_48", c->st2);
        rc = counts_class<6>(c, c->cls[1], M);
        kt_end(c, "counts_48", (c->cls[1].coo_n + c->cls[1].pend_n) * 8, c->st2);
    }
    if (!rc) rc = annotate_verb(c);
    if (rc) return rc;
    { int rj = join_st2(c); if (rj) return rj; }
    if (c->timing) HIPCHK(c, hipEventRecord(e2, c->st));
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->state.p, w_out * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    const unsigned long long* h_fb = c->h_state + w_res;
    for (int r = 0; r < MIRGE_MAX_ROUNDS; ++r) c->handed_on[r] = h_fb[r] + h_fb[w_fb + r];
    if (c->cls[0].pending || c->cls[1].pending) {
        // a fused call: the collapse was queued in front of all this and is settled only now
        bool reran = false;
        int rs = collapse_settle(c, reran);
        if (rs) return rs;
        if (reran) return annotate(c);                           // (the cascade ran over the output of an overrun partition pass: once more, over the real table)
    }
    const unsigned long long* hraw = c->h_state;
    c->h_quant.assign(hraw + w_stats, hraw + w_stats + qn); c->h_iscan.assign(hraw + w_stats + qn, hraw + w_stats + 2 * qn);
    c->h_cat.assign(hraw + w_stats + 2 * qn, hraw + w_stats + 2 * qn + w_cat);
    c->h_len.assign(hraw + w_stats + 2 * qn + w_cat, hraw + w_stats + 2 * qn + w_cat + w_len);
    c->h_tu.assign(hraw + w_res - w_tu, hraw + w_res);
    float ms = 0;
    c->annotate_ms = 0; c->counts_ms = 0;
    if (c->timing) { (void)hipEventElapsedTime(&ms, e0, e1); c->annotate_ms = ms; (void)hipEventElapsedTime(&ms, e1, e2); c->counts_ms = ms; }
    kt_resolve(c);
    uint64_t hs[MIRGE_MAX_ROUNDS * 2] = {};
    for (int r = 0; r < c->n_rounds; ++r)
        for (int w = 0; w < 2; ++w)
            for (int sl = 0; sl < STAT_SLOTS; ++sl) hs[2 * r + w] += hraw[(size_t)r * STAT_WORDS + ((size_t)w * STAT_SLOTS + sl) * 8];
#ifdef MG_PHASE_PROF
    for (int r = 0; r < c->n_rounds; ++r) {
        uint64_t a[4] = {0, 0, 0, 0};
        for (int sl = 0; sl < STAT_SLOTS; ++sl) for (int j = 0; j < 4; ++j) a[j] += hraw[(size_t)r * STAT_WORDS + (size_t)sl * 8 + 1 + j];
        fprintf(stderr, "[phase] round %d: blocks %llu  phase1 %.0f cyc/block  phase2 %.0f cyc/block  queue %.1f entries/block\n", r + 1,
                (unsigned long long)a[3], a[3] ? (double)a[0] / a[3] : 0.0, a[3] ? (double)a[1] / a[3] : 0.0, a[3] ? (double)a[2] / a[3] : 0.0);
    }
#endif
    for (int r = 0; r < c->n_rounds; ++r) {
        c->stats[r][0] = hs[2 * r]; c->stats[r][1] = hs[2 * r + 1];
        c->round_ms[r] = c->timing ? ktime(c, kRoundNames[r]).ms : 0.0;
        // algorithmic bytes of the round: 20 B per eligible unique + the 2-bit library once
        if (c->timing) ktime(c, kRoundNames[r]).bytes = 20 * hs[2 * r] + c->libs[c->rounds[r].lib].T / 4;
    }
    c->annotated = true; c->counts_valid = true; c->host_valid = false;
    return MIRGE_OK;
}

// ---- host copies for the per-unique accessors
void decode_key(const uint64_t* k, int KW, char* out, int* len_out) {
    int SW = KW == 2 ? 2 : 4;
    int len = (int)(k[SW - 1] >> 56);
    if (len > (KW == 2 ? 56 : 124)) len = KW == 2 ? 56 : 124;      // keys handed in by the caller are trusted, but never past the buffer
    for (int i = 0; i < len; ++i) {
        int code = (int)((k[i >> 5] >> (2 * (i & 31))) & 3);
        bool isN = KW == 6 && ((k[4 + (i >> 6)] >> (i & 63)) & 1);
        out[i] = isN ? 'N' : "ACGT"[code];
    }
    *len_out = len;
}

// unique i of the table -> class (0 / 1 keys, 2 verbatim) and index inside it
inline int unique_class(const mirge_ctx* c, uint64_t i, uint64_t& j) {
    const uint64_t U0 = c->cls[0].U, U1 = c->cls[1].U;
    if (i < U0) { j = i; return 0; }
    if (i < U0 + U1) { j = i - U0; return 1; }
    j = i - U0 - U1; return 2;
}
// its sequence (after fetch_host)
inline void unique_seq(const mirge_ctx* c, int k, uint64_t j, std::string& out) {
    if (k == 2) { out = c->vb.seq[(size_t)j]; return; }
    char buf[128]; int len = 0;
    decode_key(&c->h_keys[k][(size_t)j * (k ? 6 : 2)], k ? 6 : 2, buf, &len);
    out.assign(buf, (size_t)len);
}

int fetch_host(mirge_ctx* c) {
    if (c->host_valid) return MIRGE_OK;
    for (int k = 0; k < 2; ++k) {
        Class& C = c->cls[k];
        int KW = k == 0 ? 2 : 6;
        c->h_keys[k].assign((size_t)C.U * KW, 0);
        c->h_round[k].assign((size_t)C.U, 0); c->h_ref[k].assign((size_t)C.U, -1); c->h_amb[k].assign((size_t)C.U, 0);
        c->h_coo_g[k].assign((size_t)C.coo_n, 0); c->h_coo_c[k].assign((size_t)C.coo_n, 0);
        if (C.U) {
            HIPCHK(c, hipMemcpyAsync(c->h_keys[k].data(), C.ukeys.p, (size_t)C.U * KW * 8, hipMemcpyDeviceToHost, c->st));
            if (c->annotated) {
                HIPCHK(c, hipMemcpyAsync(c->h_round[k].data(), C.ann_round.p, (size_t)C.U, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, hipMemcpyAsync(c->h_ref[k].data(), C.ann_ref.p, (size_t)C.U * 4, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, hipMemcpyAsync(c->h_amb[k].data(), C.ann_amb.p, (size_t)C.U, hipMemcpyDeviceToHost, c->st));
            }
        }
        if (C.coo_n) {
            HIPCHK(c, hipMemcpyAsync(c->h_coo_g[k].data(), C.coo_g.p, (size_t)C.coo_n * 4, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipMemcpyAsync(c->h_coo_c[k].data(), C.coo_c.p, (size_t)C.coo_n * 4, hipMemcpyDeviceToHost, c->st));
        }
    }
    {
        Verb& V = c->vb;
        const size_t UV = V.seq.size();
        c->h_round[2].assign(UV, 0); c->h_ref[2].assign(UV, -1); c->h_amb[2].assign(UV, 0);
        if (UV && c->annotated) {
            HIPCHK(c, hipMemcpyAsync(c->h_round[2].data(), V.ann_round.p, UV, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipMemcpyAsync(c->h_ref[2].data(), V.ann_ref.p, UV * 4, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipMemcpyAsync(c->h_amb[2].data(), V.ann_amb.p, UV, hipMemcpyDeviceToHost, c->st));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    // CSR unique -> (sample, count); uniques are numbered [16-byte class | 48-byte class | verbatim]
    uint64_t U0 = c->cls[0].U, UK = U0 + c->cls[1].U, U = UK + c->vb.seq.size();
    c->csr_off.assign((size_t)U + 1, 0);
    for (int k = 0; k < 2; ++k) for (auto g : c->h_coo_g[k]) c->csr_off[(size_t)(g + (k ? U0 : 0)) + 1]++;
    for (auto g : c->vb.coo_u) c->csr_off[(size_t)(UK + g) + 1]++;
    for (uint64_t u = 0; u < U; ++u) c->csr_off[(size_t)u + 1] += c->csr_off[(size_t)u];
    uint64_t tot = c->csr_off[(size_t)U];
    c->csr_sample.assign((size_t)tot, 0); c->csr_count.assign((size_t)tot, 0);
    std::vector<uint64_t> cur(c->csr_off.begin(), c->csr_off.end() - 1);
    for (int k = 0; k < 2; ++k)
        for (auto& sg : c->cls[k].segs)
            for (uint64_t e = sg.start; e < sg.start + sg.count; ++e) {
                uint64_t u = c->h_coo_g[k][(size_t)e] + (k ? U0 : 0);
                uint64_t at = cur[(size_t)u]++;
                c->csr_sample[(size_t)at] = (uint32_t)sg.sample; c->csr_count[(size_t)at] = c->h_coo_c[k][(size_t)e];
            }
    for (size_t e = 0; e < c->vb.coo_u.size(); ++e) {
        uint64_t at = cur[(size_t)(UK + c->vb.coo_u[e])]++;
        c->csr_sample[(size_t)at] = c->vb.coo_s[e]; c->csr_count[(size_t)at] = c->vb.coo_c[e];
    }
    c->host_valid = true;
    return MIRGE_OK;
}

}  // namespace

// =============================================================================== C ABI

extern "C" {

const char* mirge_last_error(const mirge_ctx* c) { return c ? c->err.c_str() : g_open_error.c_str(); }

mirge_ctx* mirge_open(const char* species_dir, int32_t n_samples, int32_t device) {
    if (n_samples < 1) { fail(nullptr, MIRGE_E_ARG, "n_samples must be >= 1"); return nullptr; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fail(nullptr, MIRGE_E_HIP, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                                       " (libmirge_hip has no CPU path)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { fail(nullptr, MIRGE_E_ARG, "device index out of range"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(nullptr, MIRGE_E_HIP, "hipSetDevice failed"); return nullptr; }
    if (hipError_t ne = na_scratch_init(device); ne != hipSuccess) {
        fail(nullptr, MIRGE_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(ne)); return nullptr;
    }
    mirge_ctx* c = new mirge_ctx();
    c->device = device; c->S = n_samples;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c->n_cu = cu; }
    c->cls[0].KW = 2; c->cls[1].KW = 6;
    if (hipStreamCreate(&c->st) != hipSuccess || hipStreamCreate(&c->st2) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev_c[0]) != hipSuccess || hipEventCreate(&c->ev_c[1]) != hipSuccess || hipEventCreate(&c->ev_t[0]) != hipSuccess || hipEventCreate(&c->ev_t[1]) != hipSuccess || hipEventCreate(&c->ev_t[2]) != hipSuccess ||
        hipHostMalloc((void**)&c->h_ctr, 128 + 64) != hipSuccess) {
        fail(nullptr, MIRGE_E_HIP, "cannot create stream / events / pinned buffer"); mirge_close(c); return nullptr;
    }
    c->cls[0].st = c->st; c->cls[1].st = c->st2;
    c->cls[0].h_ctr = c->h_ctr + 16; c->cls[1].h_ctr = c->h_ctr + 20;      // inside the pinned block
    // miRge.pl:326, 333-339
    static const mirge_round def[5] = {{MIRGE_LIB_MIRNA, 26, 0, 0, 0, 28, 0, 2}, {MIRGE_LIB_HAIRPIN, 0, 25, 0, 0, 28, 1, 2},
                                       {MIRGE_LIB_OTHER, 0, 0, 0, 0, 28, 1, 2},  {MIRGE_LIB_MRNA, 0, 0, 0, 0, 28, 0, 2},
                                       {MIRGE_LIB_MIRNA, 0, 0, 1, 2, 15, 2, 2}};
    c->n_rounds = 5;
    for (int i = 0; i < 5; ++i) c->rounds[i] = def[i];
    if (const char* t = getenv("MIRGE_COOP_THR")) c->coop_thr = std::max(0, atoi(t));
    if (const char* t = getenv("MIRGE_COOP_THR3")) c->coop_thr3 = std::max(0, atoi(t));
    if (species_dir) {
        static const char* nm[MIRGE_N_LIBS] = {"mirna", "hairpin", "other", "mrna"};
        for (int l = 0; l < MIRGE_N_LIBS; ++l) {
            std::string base = std::string(species_dir) + "/" + nm[l];
            std::string path = base + ".fa";
            FILE* f = fopen(path.c_str(), "rb");
            if (!f) { path = base + ".fasta"; f = fopen(path.c_str(), "rb"); }
            if (!f) {
                // no FASTA beside the index: take the sequences out of the bowtie index itself
                FILE* e = fopen((base + ".3.ebwt").c_str(), "rb");
                if (!e) {
                    fail(nullptr, MIRGE_E_IO, "cannot find " + base + ".fa or " + base + ".{1,3,4}.ebwt");
                    mirge_close(c); return nullptr;
                }
                fclose(e);
                if (load_ebwt(c, l, base) != MIRGE_OK) { g_open_error = c->err; mirge_close(c); return nullptr; }
                continue;
            }
            fclose(f);
            if (load_fasta(c, l, path.c_str()) != MIRGE_OK) { g_open_error = c->err; mirge_close(c); return nullptr; }
        }
    }
    return c;
}

void mirge_close(mirge_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->st2) (void)hipStreamSynchronize(c->st2);
    for (auto& L : c->libs) L.release();
    for (auto& C : c->cls) C.release();
    c->vb.release();
    DevBuf* ws[] = {&c->fq_odd, &c->fq_oddtab, &c->fq_oddpool, &c->ctr, &c->stage16, &c->stage48, &c->in_seqs, &c->in_lens, &c->scan_tmp,
                    &c->fq_raw, &c->fq_seg, &c->fq_segb, &c->fq_start, &c->fq_qstart, &c->fq_qend, &c->sh_ctr, &c->sh_w[0], &c->sh_w[1], &c->hist, &c->off_tmp, &c->state};
    for (auto* b : ws) b->release();
    for (auto& k : c->kt) { if (k.a) (void)hipEventDestroy(k.a); if (k.b) (void)hipEventDestroy(k.b); }
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->h_odd) (void)hipHostFree(c->h_odd);
    for (auto& e : c->ev_t) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_c) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->st2) (void)hipStreamDestroy(c->st2);
    if (c->fq_pin) (void)hipHostFree(c->fq_pin);
    if (c->st) (void)hipStreamDestroy(c->st);
    delete c;
}

int32_t mirge_reset(mirge_ctx* c, int32_t n_samples) {
    if (!c) return MIRGE_E_ARG;
    if (n_samples < 1) return fail(c, MIRGE_E_ARG, "n_samples must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    c->S = n_samples;
    for (auto& C : c->cls) C.reset();
    c->vb.reset();
    c->annotated = false; c->counts_valid = false; c->host_valid = false;
    return MIRGE_OK;
}

int32_t mirge_load_library_ebwt(mirge_ctx* c, int32_t lib, const char* ebwt_base) {
    if (!c || !ebwt_base) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return load_ebwt(c, lib, ebwt_base);
}

int32_t mirge_load_library_fasta(mirge_ctx* c, int32_t lib, const char* path) {
    if (!c || !path) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return load_fasta(c, lib, path);
}

int32_t mirge_load_library_mem(mirge_ctx* c, int32_t lib, int32_t n, const char* names, const char* seqs, const uint64_t* offs) {
    if (!c || n < 0 || (n > 0 && (!names || !seqs)) || !offs) return c ? fail(c, MIRGE_E_ARG, "bad library arguments") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<std::string> nm; nm.reserve((size_t)n);
    const char* p = names;
    for (int i = 0; i < n; ++i) { nm.emplace_back(p); p += nm.back().size() + 1; }
    return load_library(c, lib, std::move(nm), seqs, offs);
}

int32_t mirge_lib_size(const mirge_ctx* c, int32_t lib) {
    if (!c || lib < 0 || lib >= MIRGE_N_LIBS || !c->libs[lib].loaded) return MIRGE_E_ARG;
    return (int32_t)c->libs[lib].names.size();
}
const char* mirge_lib_name(const mirge_ctx* c, int32_t lib, int32_t idx) {
    if (!c || lib < 0 || lib >= MIRGE_N_LIBS || idx < 0 || (size_t)idx >= c->libs[lib].names.size()) return nullptr;
    return c->libs[lib].names[(size_t)idx].c_str();
}
const char* mirge_lib_short_name(const mirge_ctx* c, int32_t lib, int32_t idx) {
    if (!c || lib < 0 || lib >= MIRGE_N_LIBS || idx < 0 || (size_t)idx >= c->libs[lib].short_names.size()) return nullptr;
    return c->libs[lib].short_names[(size_t)idx].c_str();
}

int32_t mirge_set_rounds(mirge_ctx* c, int32_t n, const mirge_round* r) {
    if (!c) return MIRGE_E_ARG;
    if (n < 1 || n > MIRGE_MAX_ROUNDS || !r) return fail(c, MIRGE_E_ARG, "between 1 and 8 rounds");
    for (int i = 0; i < n; ++i) {
        if (r[i].lib < 0 || r[i].lib >= MIRGE_N_LIBS || r[i].trim5 < 0 || r[i].trim3 < 0 || r[i].seedlen < 1 ||
            r[i].seedmm < 0 || r[i].seedmm > 3 || r[i].maxmm < 0 || r[i].maxmm > 3)
            return fail(c, MIRGE_E_ARG, "round " + std::to_string(i + 1) + ": parameter out of range");
        c->rounds[i] = r[i];
    }
    c->n_rounds = n; c->annotated = false; c->counts_valid = false;
    return MIRGE_OK;
}

int32_t mirge_collapse_seqs(mirge_ctx* c, const char* seqs, const uint64_t* offs, uint64_t n, int32_t sample) {
    if (!c || !offs || (n && !seqs)) return c ? fail(c, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return collapse_host(c, seqs, offs, n, sample);
}

int32_t mirge_collapse_fastq(mirge_ctx* c, const char* path, int32_t sample, uint64_t* n_reads) {
    if (!c || !path) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t size = 0; uint8_t last = '\n';
    int rc = fastq_file_to_hbm(c, path, &size, &last);
    if (rc) return rc;
    return collapse_fastq_dev(c, c->fq_raw.as<uint8_t>(), size, last, sample, n_reads);
}

int32_t mirge_collapse_fastq_mem(mirge_ctx* c, const char* data, uint64_t size, int32_t sample, uint64_t* n_reads) {
    if (!c || (size && !data)) return c ? fail(c, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->fq_raw.ensure(size + 256, false, 0, c->st));
    if (size) HIPCHK(c, hipMemcpyAsync(c->fq_raw.p, data, size, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return collapse_fastq_dev(c, c->fq_raw.as<uint8_t>(), size, size ? (uint8_t)data[size - 1] : (uint8_t)'\n', sample, n_reads);
}

int32_t mirge_collapse_fastq_dev(mirge_ctx* c, const void* d_data, uint64_t size, int32_t sample, uint64_t* n_reads) {
    if (!c || (size && !d_data)) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t last = '\n';
    if (size) {
        HIPCHK(c, hipMemcpyAsync(c->h_ctr + 9, (const uint8_t*)d_data + size - 1, 1, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        last = *(const uint8_t*)(c->h_ctr + 9);
    }
    return collapse_fastq_dev(c, (const uint8_t*)d_data, size, last, sample, n_reads);
}

int32_t mirge_collapse_raw_fastq(mirge_ctx* c, const char* path, int32_t sample, const char* adapter, int32_t trim5, int32_t trim3,
                                 uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred) {
    if (!c || !path) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    TrimPlan tp;
    int rc = make_trim_plan(c, adapter, trim5, trim3, tp);
    if (rc) return rc;
    uint64_t size = 0; uint8_t last = '\n';
    rc = fastq_file_to_hbm(c, path, &size, &last);
    if (rc) return rc;
    rc = collapse_fastq_dev(c, c->fq_raw.as<uint8_t>(), size, last, sample, total_reads, &tp);
    if (trimmed_reads) *trimmed_reads = tp.kept;
    if (phred) *phred = tp.phred_reported;
    return rc;
}

int32_t mirge_collapse_raw_fastq_mem(mirge_ctx* c, const char* data, uint64_t size, int32_t sample, const char* adapter, int32_t trim5,
                                     int32_t trim3, uint64_t* total_reads, uint64_t* trimmed_reads, int32_t* phred) {
    if (!c || (size && !data)) return c ? fail(c, MIRGE_E_ARG, "null input") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    TrimPlan tp;
    int rc = make_trim_plan(c, adapter, trim5, trim3, tp);
    if (rc) return rc;
    HIPCHK(c, c->fq_raw.ensure(size + 256, false, 0, c->st));
    if (size) HIPCHK(c, hipMemcpyAsync(c->fq_raw.p, data, size, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    rc = collapse_fastq_dev(c, c->fq_raw.as<uint8_t>(), size, size ? (uint8_t)data[size - 1] : (uint8_t)'\n', sample, total_reads, &tp);
    if (trimmed_reads) *trimmed_reads = tp.kept;
    if (phred) *phred = tp.phred_reported;
    return rc;
}

int32_t mirge_shard_export_sizes(mirge_ctx* c, int32_t world, uint64_t* n16_per_rank, uint64_t* n48_per_rank) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return shard_export_sizes(c, world, n16_per_rank, n48_per_rank);
}
int32_t mirge_shard_export(mirge_ctx* c, int32_t world, void* d_rec16, void* d_rec48) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return shard_export(c, world, d_rec16, d_rec48);
}
int32_t mirge_shard_import(mirge_ctx* c, const void* d_rec16, uint64_t n16, const void* d_rec48, uint64_t n48) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return shard_import(c, d_rec16, n16, d_rec48, n48);
}

// verbatim uniques across ranks: a byte blob of [len u32 | sample u32 | count u32 | len bytes] per entry, host memory
uint64_t mirge_verb_export_size(const mirge_ctx* c) {
    if (!c) return 0;
    uint64_t n = 0;
    for (size_t e = 0; e < c->vb.coo_u.size(); ++e) n += 12 + c->vb.seq[c->vb.coo_u[e]].size();
    return n;
}
int32_t mirge_verb_export(mirge_ctx* c, void* blob) {
    if (!c || (!blob && mirge_verb_export_size(c))) return c ? fail(c, MIRGE_E_ARG, "null buffer") : MIRGE_E_ARG;
    uint8_t* o = (uint8_t*)blob;
    for (size_t e = 0; e < c->vb.coo_u.size(); ++e) {
        const std::string& q = c->vb.seq[c->vb.coo_u[e]];
        const uint32_t h[3] = {(uint32_t)q.size(), c->vb.coo_s[e], c->vb.coo_c[e]};
        memcpy(o, h, 12); memcpy(o + 12, q.data(), q.size()); o += 12 + q.size();
    }
    return MIRGE_OK;
}
int32_t mirge_verb_clear(mirge_ctx* c) {
    if (!c) return MIRGE_E_ARG;
    c->vb.reset(); c->annotated = false; c->counts_valid = false; c->host_valid = false;
    return MIRGE_OK;
}
int32_t mirge_verb_import(mirge_ctx* c, const void* blob, uint64_t nbytes, int32_t rank, int32_t world) {
    if (!c || (nbytes && !blob) || world < 1 || rank < 0 || rank >= world) return c ? fail(c, MIRGE_E_ARG, "bad arguments") : MIRGE_E_ARG;
    const uint8_t* p = (const uint8_t*)blob; const uint8_t* end = p + nbytes;
    while (p < end) {
        uint32_t h[3];
        if (end - p < 12) return fail(c, MIRGE_E_FORMAT, "truncated verbatim-read blob");
        memcpy(h, p, 12); p += 12;
        if ((uint64_t)(end - p) < h[0] || h[0] > MIRGE_MAX_READ_LEN || h[1] >= (uint32_t)c->S || h[2] == 0) return fail(c, MIRGE_E_FORMAT, "corrupt verbatim-read blob");
        uint64_t f = 1469598103934665603ull;
        for (uint32_t i = 0; i < h[0]; ++i) { f ^= p[i]; f *= 1099511628211ull; }
        if ((int32_t)((f >> 17) % (uint64_t)world) == rank) c->vb.add((const char*)p, h[0], (int)h[1], c->S, h[2]);
        p += h[0];
    }
    c->annotated = false; c->counts_valid = false; c->host_valid = false;
    return MIRGE_OK;
}

int32_t mirge_pack_dev(mirge_ctx* c, const void* d_seqs, uint32_t width, const void* d_lens, uint64_t n,
                       void* d_keys16, uint64_t* n16, void* d_keys48, uint64_t* n48) {
    if (!c || (n && (!d_seqs || !d_lens || !d_keys16 || !d_keys48))) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->ctr.ensure(128, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->ctr.p, 0, 128, c->st));
    c->h_ctr[0] = c->h_ctr[1] = 0;
    int rc = pack_launch(c, d_seqs, width, d_lens, n, d_keys16, d_keys48);
    if (rc) return rc;
    if (n16) *n16 = n ? c->h_ctr[0] : 0;
    if (n48) *n48 = n ? c->h_ctr[1] : 0;
    kt_resolve(c);
    return MIRGE_OK;
}

int32_t mirge_collapse_ascii_dev(mirge_ctx* c, const void* d_seqs, uint32_t width, const void* d_lens, uint64_t n, int32_t sample) {
    if (!c || (n && (!d_seqs || !d_lens))) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->stage16.ensure(std::max<uint64_t>(n, 1) * 16, false, 0, c->st));
    HIPCHK(c, c->stage48.ensure(std::max<uint64_t>(n, 1) * 48, false, 0, c->st));
    uint64_t n16 = 0, n48 = 0;
    int rc = mirge_pack_dev(c, d_seqs, width, d_lens, n, c->stage16.p, &n16, c->stage48.p, &n48);
    if (rc) return rc;
    return collapse_packed(c, c->stage16.p, n16, c->stage48.p, n48, sample);
}

// collapse + cascade + count vectors of one sample in one call, with nothing waited for in between
int32_t mirge_collapse_annotate_packed_dev(mirge_ctx* c, const void* d16, uint64_t n16, const void* d48, uint64_t n48, int32_t sample) {
    if (!c || (n16 && !d16) || (n48 && !d48)) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    static const bool defer = [] { const char* e = getenv("MIRGE_DEFER"); return !(e && atoi(e) == 0); }();      // A/B aid
    int rc = collapse_packed(c, d16, n16, d48, n48, sample, defer);
    if (!rc) rc = annotate(c);
    if (rc && (c->cls[0].pending || c->cls[1].pending)) {        // the cascade could not be queued: settle the collapse alone
        (void)hipStreamSynchronize(c->st); (void)hipStreamSynchronize(c->st2);
        bool reran = false; (void)collapse_settle(c, reran);
    }
    return rc;
}

int32_t mirge_validate_keys_dev(mirge_ctx* c, const void* d16, uint64_t n16, const void* d48, uint64_t n48) {
    if (!c || (n16 && !d16) || (n48 && !d48)) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->ctr.ensure(128, false, 0, c->st));
    HIPCHK(c, hipMemsetAsync(c->ctr.p, 0xFF, 16, c->st));
    unsigned long long* f = c->ctr.as<unsigned long long>();
    if (n16) hipLaunchKernelGGL((k_validate_keys<2>), dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, c->st, (const uint64_t*)d16, n16, f);
    if (n48) hipLaunchKernelGGL((k_validate_keys<6>), dim3((uint32_t)((n48 + 255) / 256)), dim3(256), 0, c->st, (const uint64_t*)d48, n48, f + 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->ctr.p, 16, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (c->h_ctr[0] != ~0ull) return fail(c, MIRGE_E_FORMAT, "16-byte key " + std::to_string(c->h_ctr[0] - 1) + " is malformed (length byte, bits above the sequence)");
    if (c->h_ctr[1] != ~0ull) return fail(c, MIRGE_E_FORMAT, "48-byte key " + std::to_string(c->h_ctr[1] - 1) + " is malformed (length byte, bits above the sequence, N mask)");
    return MIRGE_OK;
}

int32_t mirge_collapse_packed_dev(mirge_ctx* c, const void* d16, uint64_t n16, const void* d48, uint64_t n48, int32_t sample) {
    if (!c || (n16 && !d16) || (n48 && !d48)) return c ? fail(c, MIRGE_E_ARG, "null device pointer") : MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return collapse_packed(c, d16, n16, d48, n48, sample);
}

// The collapse of ONE stream of 16-byte keys with nothing allocated: the kernels of mirge_collapse_packed_dev, their
// workspaces in this GPU's copy of `scratch`, the results in the caller's buffers.  No hipMalloc / hipFree / event or
// stream creation on this path (the stream and the pinned word it reads the result through were made by mirge_open).
int32_t mirge_collapse_keys16_noalloc(mirge_ctx* c, const void* d_keys16, uint64_t n, void* d_ukeys, uint32_t* d_counts, uint64_t* n_unique) {
    if (!c) return MIRGE_E_ARG;
    if (!n_unique || (n && (!d_keys16 || !d_ukeys || !d_counts))) return fail(c, MIRGE_E_ARG, "null pointer");
    if (n > NA_MAX_KEYS)
        return fail(c, MIRGE_E_LIMIT, "mirge_collapse_keys16_noalloc takes at most " + std::to_string(NA_MAX_KEYS) + " keys per call (" +
                                          std::to_string(n) + " given): its workspace is a fixed array");
    if (((uintptr_t)d_keys16 | (uintptr_t)d_ukeys) & 7) return fail(c, MIRGE_E_ARG, "key buffers must be 8-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    NaLayout* W = (c->device >= 0 && c->device < NA_MAX_DEVICES) ? g_na_ptr[c->device] : nullptr;
    if (!W) return fail(c, MIRGE_E_HIP, "no workspace address for this device");
    *n_unique = 0;
    if (n == 0) return MIRGE_OK;
    const uint32_t n32 = (uint32_t)n;
    const uint64_t cap = pow2_at_least(n + n / 4);              // <= NA_SLOTS
    // the array keeps the last call's contents: everything that must start out empty / zero is cleared here
    HIPCHK(c, hipMemsetAsync(W->lslots, 0xFF, cap * 4, c->st));
    HIPCHK(c, hipMemsetAsync(W->cnt, 0, n * 4, c->st));
    HIPCHK(c, hipMemsetAsync(W->ctr, 0, sizeof W->ctr, c->st));
    constexpr int T = MG_CL_T, CB = MG_CL_B;
    hipLaunchKernelGGL((k_collapse_local<2, T, CB>), dim3((n32 + T - 1) / T), dim3(CB), 0, c->st, (const uint64_t*)d_keys16,
                       (const uint32_t*)nullptr, n32, W->lslots, (uint32_t)(cap - 1), W->cnt, W->cnt0);
    hipLaunchKernelGGL((k_collapse_emit<2>), dim3((n32 + 256 * EMIT_PER - 1) / (256 * EMIT_PER)), dim3(256), 0, c->st,
                       (const uint64_t*)d_keys16, W->cnt, W->cnt0, n32, (const uint32_t*)nullptr, 0u, (uint64_t*)d_ukeys, 0u,
                       W->coo_g, d_counts, W->ctr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_ctr + 10, W->ctr, 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    // no earlier table: every representative is a new unique, so entry j of the counts belongs to unique j
    *n_unique = c->h_ctr[10] >> 32;
    return MIRGE_OK;
}

int32_t mirge_annotate(mirge_ctx* c) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return annotate(c);
}

int32_t mirge_round_stats(const mirge_ctx* c, int32_t round, uint64_t* processed, uint64_t* aligned, double* seconds) {
    if (!c || round < 1 || round > c->n_rounds || !c->annotated) return MIRGE_E_ARG;
    if (processed) *processed = c->stats[round - 1][0];
    if (aligned) *aligned = c->stats[round - 1][1];
    if (seconds) *seconds = c->round_ms[round - 1] * 1e-3;
    return MIRGE_OK;
}

int32_t mirge_round_handed_on(const mirge_ctx* c, int32_t round, uint64_t* n) {
    if (!c || round < 1 || round > c->n_rounds || !c->annotated || !n) return MIRGE_E_ARG;
    *n = c->handed_on[round - 1];
    return MIRGE_OK;
}

int32_t mirge_counts(mirge_ctx* c, const uint64_t** quant, const uint64_t** iscan, const uint64_t** category,
                     const uint64_t** len_hist, const uint64_t** trimmed_uniq) {
    if (!c) return MIRGE_E_ARG;
    if (!c->counts_valid) return fail(c, MIRGE_E_ARG, "mirge_counts before mirge_annotate");
    if (quant) *quant = c->h_quant.data();
    if (iscan) *iscan = c->h_iscan.data();
    if (category) *category = c->h_cat.data();
    if (len_hist) *len_hist = c->h_len.data();
    if (trimmed_uniq) *trimmed_uniq = c->h_tu.data();
    return MIRGE_OK;
}

uint64_t mirge_counts_flat_len(const mirge_ctx* c) {
    if (!c || !c->libs[MIRGE_LIB_MIRNA].loaded) return 0;
    uint64_t M = c->libs[MIRGE_LIB_MIRNA].names.size(), S = (uint64_t)c->S;
    return 2 * M * S + 5 * S + (MIRGE_MAX_READ_LEN + 1) * S + S + 2 * MIRGE_MAX_ROUNDS;
}
int32_t mirge_counts_flat_get(mirge_ctx* c, uint64_t* out) {
    if (!c || !out) return MIRGE_E_ARG;
    if (!c->counts_valid) return fail(c, MIRGE_E_ARG, "count vectors not computed yet");
    size_t M = c->libs[MIRGE_LIB_MIRNA].names.size(), S = (size_t)c->S, o = 0;
    memcpy(out + o, c->h_quant.data(), M * S * 8); o += M * S;
    memcpy(out + o, c->h_iscan.data(), M * S * 8); o += M * S;
    memcpy(out + o, c->h_cat.data(), 5 * S * 8); o += 5 * S;
    memcpy(out + o, c->h_len.data(), (MIRGE_MAX_READ_LEN + 1) * S * 8); o += (MIRGE_MAX_READ_LEN + 1) * S;
    memcpy(out + o, c->h_tu.data(), S * 8); o += S;
    for (int r = 0; r < MIRGE_MAX_ROUNDS; ++r) { out[o + r] = c->stats[r][0]; out[o + MIRGE_MAX_ROUNDS + r] = c->stats[r][1]; }
    return MIRGE_OK;
}
int32_t mirge_counts_flat_set(mirge_ctx* c, const uint64_t* in) {
    if (!c || !in) return MIRGE_E_ARG;
    if (!c->counts_valid) return fail(c, MIRGE_E_ARG, "count vectors not computed yet");
    size_t M = c->libs[MIRGE_LIB_MIRNA].names.size(), S = (size_t)c->S, o = 0;
    memcpy(c->h_quant.data(), in + o, M * S * 8); o += M * S;
    memcpy(c->h_iscan.data(), in + o, M * S * 8); o += M * S;
    memcpy(c->h_cat.data(), in + o, 5 * S * 8); o += 5 * S;
    memcpy(c->h_len.data(), in + o, (MIRGE_MAX_READ_LEN + 1) * S * 8); o += (MIRGE_MAX_READ_LEN + 1) * S;
    memcpy(c->h_tu.data(), in + o, S * 8); o += S;
    for (int r = 0; r < MIRGE_MAX_ROUNDS; ++r) { c->stats[r][0] = in[o + r]; c->stats[r][1] = in[o + MIRGE_MAX_ROUNDS + r]; }
    return MIRGE_OK;
}

// the same buffer on the device (for the all-reduce: RCCL reduces device memory, no bounce through the host)
__global__ void __launch_bounds__(256)
k_flat_pack(const unsigned long long* __restrict__ arena, uint64_t w_stats, uint64_t qn2, uint64_t w_cat, uint64_t S, int n_rounds,
            unsigned long long* __restrict__ out) {
    // arena: [round stats | quant | iscan | categories | LEN_BINS x S length histogram | trimmedUniq]; out: the flat layout
    const uint64_t n_len = (uint64_t)(MIRGE_MAX_READ_LEN + 1) * S, head = qn2 + w_cat;
    const uint64_t total = head + n_len + S + 2 * MIRGE_MAX_ROUNDS;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long v = 0;
        if (i < head + n_len + S) v = arena[w_stats + i];
        else {
            const uint64_t j = i - head - n_len - S;                 // j < MAX_ROUNDS: processed, else aligned
            const int r = (int)(j % MIRGE_MAX_ROUNDS), w = (int)(j / MIRGE_MAX_ROUNDS);
            if (r < n_rounds) for (int sl = 0; sl < STAT_SLOTS; ++sl) v += arena[(size_t)r * STAT_WORDS + ((size_t)w * STAT_SLOTS + sl) * 8];
        }
        out[i] = v;
    }
}
int32_t mirge_counts_flat_get_dev(mirge_ctx* c, void* d_out) {
    if (!c || !d_out) return MIRGE_E_ARG;
    if (!c->counts_valid || !c->state.p) return fail(c, MIRGE_E_ARG, "count vectors not computed yet");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t M = c->libs[MIRGE_LIB_MIRNA].names.size(), S = (uint64_t)c->S, qn = std::max<uint64_t>(M * S, 1);
    if (M * S == 0) return fail(c, MIRGE_E_ARG, "empty miRNA library");
    hipLaunchKernelGGL(k_flat_pack, dim3(64), dim3(256), 0, c->st, c->state.as<unsigned long long>(), (uint64_t)(MIRGE_MAX_ROUNDS * STAT_WORDS),
                       2 * qn, 5 * S, S, c->n_rounds, (unsigned long long*)d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->st));
    return MIRGE_OK;
}
int32_t mirge_counts_flat_set_dev(mirge_ctx* c, const void* d_in) {
    if (!c || !d_in) return MIRGE_E_ARG;
    if (!c->counts_valid) return fail(c, MIRGE_E_ARG, "count vectors not computed yet");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> h((size_t)mirge_counts_flat_len(c));
    HIPCHK(c, hipMemcpyAsync(h.data(), d_in, h.size() * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return mirge_counts_flat_set(c, h.data());
}

uint64_t mirge_n_unique(const mirge_ctx* c) { return c ? c->cls[0].U + c->cls[1].U + c->vb.seq.size() : 0; }
uint32_t mirge_max_unique_len(const mirge_ctx* c) {
    if (!c) return 0;
    return std::max<uint32_t>(c->vb.maxlen, c->cls[1].U ? (uint32_t)KeyTraits<6>::MAXLEN : (uint32_t)KeyTraits<2>::MAXLEN);
}

int32_t mirge_export_uniques(mirge_ctx* c, char* seqs, uint32_t stride, uint16_t* lens, int8_t* ann_round, int32_t* ann_ref, uint8_t* ann_amb) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fetch_host(c);
    if (rc) return rc;
    const uint64_t U = mirge_n_unique(c);
    std::string sq;
    for (uint64_t u = 0; u < U; ++u) {
        uint64_t i; const int k = unique_class(c, u, i);
        if (seqs || lens) unique_seq(c, k, i, sq);
        if (seqs) {
            if (sq.size() > stride) return fail(c, MIRGE_E_ARG, "stride smaller than a unique sequence (mirge_max_unique_len)");
            memset(seqs + u * stride, 0, stride); memcpy(seqs + u * stride, sq.data(), sq.size());
        }
        if (lens) lens[u] = (uint16_t)sq.size();
        if (ann_round) ann_round[u] = (int8_t)c->h_round[k][(size_t)i];
        if (ann_ref) ann_ref[u] = c->h_ref[k][(size_t)i];
        if (ann_amb) ann_amb[u] = c->h_amb[k][(size_t)i];
    }
    return MIRGE_OK;
}

int32_t mirge_unique(mirge_ctx* c, uint64_t i, const char** seq, int32_t ref_idx[MIRGE_MAX_ROUNDS], uint8_t* ambiguous, const uint32_t** counts) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fetch_host(c);
    if (rc) return rc;
    if (i >= mirge_n_unique(c)) return fail(c, MIRGE_E_ARG, "unique index out of range");
    uint64_t j; const int k = unique_class(c, i, j);
    unique_seq(c, k, j, c->uniq_seq);
    if (seq) *seq = c->uniq_seq.c_str();
    int r = c->h_round[k][(size_t)j];
    if (ref_idx) { for (int q = 0; q < MIRGE_MAX_ROUNDS; ++q) ref_idx[q] = -1; if (r > 0) ref_idx[r - 1] = c->h_ref[k][(size_t)j]; }
    if (ambiguous) *ambiguous = (r > 0 && c->h_amb[k][(size_t)j]) ? (uint8_t)(1u << (r - 1)) : 0;
    if (counts) {
        c->uniq_counts.assign((size_t)c->S, 0);
        for (uint64_t e = c->csr_off[(size_t)i]; e < c->csr_off[(size_t)i + 1]; ++e) c->uniq_counts[c->csr_sample[(size_t)e]] += c->csr_count[(size_t)e];
        *counts = c->uniq_counts.data();
    }
    return MIRGE_OK;
}

// ---- the host tail of the run in native code: summarize's bookkeeping, miRNAmerge, filter and the four
// tables of writeDataToCSV, for a batch mode without Perl.  Same steps, same quirks, as the reference's subs.
int32_t mirge_write_tables(mirge_ctx* c, const char* merges_csv, const char* output_dir, const char* const* sample_names,
                           uint64_t* mirna_reads_filtered, uint64_t* mirna_uniq_filtered) {
    if (!c || !output_dir || !sample_names) return c ? fail(c, MIRGE_E_ARG, "null argument") : MIRGE_E_ARG;
    if (!c->annotated || !c->counts_valid) return fail(c, MIRGE_E_ARG, "mirge_annotate has not run");
    if (c->n_rounds != 5) return fail(c, MIRGE_E_ARG, "the tables are those of the five-round cascade (miRge.pl:111)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fetch_host(c);
    if (rc) return rc;
    const int S = c->S;
    const Lib& mir = c->libs[MIRGE_LIB_MIRNA];
    const size_t M = mir.names.size();
    struct Ent { std::vector<uint64_t> quant, iscan; };
    std::map<std::string, Ent> mirHash;                        // std::map orders bytewise: the `sort keys` of miRge.pl:878, 894
    auto entry = [&](const std::string& k) -> Ent& {
        Ent& e = mirHash[k];
        if (e.quant.empty()) { e.quant.assign((size_t)S, 0); e.iscan.assign((size_t)S, 0); }
        return e;
    };
    // summarize, miRge.pl:457-463 (keys = what bowtie-inspect -n prints) and :473-476 (counts go to the name bowtie reports)
    for (size_t m = 0; m < M; ++m) entry(mir.names[m]);
    for (size_t m = 0; m < M; ++m)
        for (int s = 0; s < S; ++s) {
            const uint64_t q = c->h_quant[m * S + s], i = c->h_iscan[m * S + s];
            if (q || i) { Ent& e = entry(mir.short_names[m]); e.quant[(size_t)s] += q; e.iscan[(size_t)s] += i; }
        }
    // miRNAmerge, miRge.pl:496-524
    if (merges_csv) {
        std::string data;
        FILE* f = fopen(merges_csv, "rb");
        if (f) {
            fclose(f);
            rc = read_file(c, merges_csv, data);
            if (rc) return rc;
            size_t i = 0;
            while (i < data.size()) {
                size_t e = data.find('\n', i); if (e == std::string::npos) e = data.size();
                std::vector<std::string> cols;                  // split(',', $line): trailing empty fields dropped; chomp strips only "\n"
                size_t a = i;
                for (;;) { size_t b = data.find(',', a); if (b == std::string::npos || b > e) b = e; cols.emplace_back(data, a, b - a); if (b >= e) break; a = b + 1; }
                while (!cols.empty() && cols.back().empty()) cols.pop_back();
                for (size_t k = 1; k < cols.size(); ++k) {
                    auto it = mirHash.find(cols[k]);
                    if (it == mirHash.end()) continue;
                    for (int j = 0; j < S; ++j)
                        if (it->second.quant[(size_t)j] > 0) {                                  // :511
                            Ent& t = entry(cols[0]);
                            const uint64_t q = it->second.quant[(size_t)j], sc = it->second.iscan[(size_t)j];
                            t.quant[(size_t)j] += q; t.iscan[(size_t)j] += sc;                  // :512-513
                        }
                    mirHash.erase(cols[k]);                                                      // :516
                }
                i = e + 1;
            }
        }                                                       // a missing merges file is not an error (:521-523)
    }
    // filter, miRge.pl:526-554
    std::vector<uint64_t> tot((size_t)S, 0), uq((size_t)S, 0);
    for (auto& kv : mirHash) for (int i = 0; i < S; ++i) if (kv.second.iscan[(size_t)i] < 2) kv.second.quant[(size_t)i] = 0;
    for (auto& kv : mirHash) for (int i = 0; i < S; ++i) if (kv.second.quant[(size_t)i] > 0) { tot[(size_t)i] += kv.second.quant[(size_t)i]; uq[(size_t)i] += 1; }
    for (int i = 0; i < S; ++i) { if (mirna_reads_filtered) mirna_reads_filtered[i] = tot[(size_t)i]; if (mirna_uniq_filtered) mirna_uniq_filtered[i] = uq[(size_t)i]; }
    // writeDataToCSV, miRge.pl:690-907
    const std::string dir = output_dir;
    (void)mkdir(dir.c_str(), 0777);
    std::string head = "uniqueSequence,annotFlag,exact miRNA,hairpin miRNA,non miRNA/mRNA RNA,mRNA,isomiR miRNA", names;
    for (int i = 0; i < S; ++i) { names += ","; names += sample_names[i] ? sample_names[i] : ""; }
    head += names + "\n";
    FILE* fm = fopen((dir + "/mapped.csv").c_str(), "wb");
    FILE* fu = fopen((dir + "/unmapped.csv").c_str(), "wb");
    FILE* fc = fopen((dir + "/miR.Counts.csv").c_str(), "wb");
    FILE* fr = fopen((dir + "/miR.RPM.csv").c_str(), "wb");
    if (!fm || !fu || !fc || !fr) {
        for (FILE* f : {fm, fu, fc, fr}) if (f) fclose(f);
        return fail(c, MIRGE_E_IO, "cannot write the tables into " + dir);
    }
    fputs(head.c_str(), fm); fputs(head.c_str(), fu);
    const uint64_t U = mirge_n_unique(c);
    std::vector<uint64_t> cnt((size_t)S);
    std::string row;
    for (uint64_t u = 0; u < U; ++u) {
        uint64_t j; const int k = unique_class(c, u, j);
        const int r = c->h_round[k][(size_t)j];
        std::fill(cnt.begin(), cnt.end(), 0);
        for (uint64_t e = c->csr_off[(size_t)u]; e < c->csr_off[(size_t)u + 1]; ++e) cnt[c->csr_sample[(size_t)e]] += c->csr_count[(size_t)e];
        unique_seq(c, k, j, row);
        row += r > 0 ? ",1" : ",0";
        for (int q = 1; q <= 5; ++q) { row += ","; if (q == r) row += c->libs[c->rounds[q - 1].lib].short_names[(size_t)c->h_ref[k][(size_t)j]]; }
        for (int i = 0; i < S; ++i) { row += ","; row += std::to_string(cnt[(size_t)i]); }
        row += r > 0 ? ",\n" : "\n";                            // mapped rows end ",\n" (:748-749), unmapped rows do not (:853-860)
        fputs(row.c_str(), r > 0 ? fm : fu);
    }
    std::string h2 = "miRNA" + names + "\n";
    fputs(h2.c_str(), fc); fputs(h2.c_str(), fr);
    row = "miRNAtotal";
    for (int i = 0; i < S; ++i) { row += ","; if (uq[(size_t)i]) row += std::to_string(tot[(size_t)i]); }      // undef => empty cell (:874)
    row += "\n"; fputs(row.c_str(), fc);
    for (auto& kv : mirHash) {
        row = kv.first;
        for (int i = 0; i < S; ++i) { row += ","; row += std::to_string(kv.second.quant[(size_t)i]); }
        row += "\n"; fputs(row.c_str(), fc);
        row = kv.first;
        for (int i = 0; i < S; ++i) {
            char num[64] = "0";
            if (tot[(size_t)i]) snprintf(num, sizeof num, "%.15g", (double)(1000000ull * kv.second.quant[(size_t)i]) / (double)tot[(size_t)i]);   // :897-898
            row += ","; row += num;
        }
        row += "\n"; fputs(row.c_str(), fr);
    }
    bool bad = false;
    for (FILE* f : {fm, fu, fc, fr}) bad |= fclose(f) != 0;
    return bad ? fail(c, MIRGE_E_IO, "short write in " + dir) : MIRGE_OK;
}

uint64_t mirge_n_counts(const mirge_ctx* c) { return c ? c->cls[0].coo_n + c->cls[1].coo_n + c->vb.coo_u.size() : 0; }

int32_t mirge_export_counts(mirge_ctx* c, uint64_t* uid, uint32_t* sample, uint32_t* count) {
    if (!c) return MIRGE_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = fetch_host(c);
    if (rc) return rc;
    uint64_t U = mirge_n_unique(c), o = 0;
    for (uint64_t u = 0; u < U; ++u)
        for (uint64_t e = c->csr_off[(size_t)u]; e < c->csr_off[(size_t)u + 1]; ++e, ++o) {
            if (uid) uid[o] = u;
            if (sample) sample[o] = c->csr_sample[(size_t)e];
            if (count) count[o] = c->csr_count[(size_t)e];
        }
    return MIRGE_OK;
}

int32_t mirge_last_timing(const mirge_ctx* c, double* collapse_ms, double* annotate_ms, double* counts_ms) {
    if (!c) return MIRGE_E_ARG;
    if (collapse_ms) *collapse_ms = c->collapse_ms;
    if (annotate_ms) *annotate_ms = c->annotate_ms;
    if (counts_ms) *counts_ms = c->counts_ms;
    return MIRGE_OK;
}

int32_t mirge_set_kernel_timing(mirge_ctx* c, int32_t on) {
    if (!c) return MIRGE_E_ARG;
    c->timing = on != 0;
    if (!c->timing) for (auto& k : c->kt) { k.used = false; k.ms = 0; }
    return MIRGE_OK;
}

int32_t mirge_kernel_times(const mirge_ctx* c, const char** names, double* ms, uint64_t* bytes_alg, int32_t cap) {
    if (!c) return MIRGE_E_ARG;
    int32_t n = 0;
    for (auto& k : c->kt) {
        if (!k.used) continue;
        if (n < cap) { if (names) names[n] = k.name; if (ms) ms[n] = k.ms; if (bytes_alg) bytes_alg[n] = k.bytes; }
        ++n;
    }
    return n;
}

}  // extern "C"

#include "group.h"
