/*
 * mirge_oracle.c -- CPU restatement of miRge's collapse + library cascade.
 * TEST INFRASTRUCTURE ONLY (see mirge_oracle.h for scope and parity status).
 *
 * Works on ASCII on purpose: it shares no packing, hashing or bit tricks with
 * the HIP implementation it checks.
 */
#include "mirge_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ library */

struct orc_lib {
    int32_t   n;
    char*     text;     /* concatenated, upper-cased, non-ACGT -> 'N' */
    uint64_t* offs;     /* n+1 */
    /* indexed matcher: CSR over the kb-mer at every position of every reference */
    int32_t   kb;
    uint32_t* koff;     /* 4^kb + 1 */
    uint32_t* kref;     /* reference index of each occurrence */
    uint32_t* kpos;     /* offset inside that reference       */
};

static inline int code_of(char c) {
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}
static inline char comp_of(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}
static void lib_build_index(orc_lib* L) {
    int kb = L->kb;
    uint64_t nb = 1ull << (2 * kb);
    L->koff = (uint32_t*)calloc(nb + 1, sizeof(uint32_t));
    uint64_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int32_t r = 0; r < L->n; ++r) {
            const char* s = L->text + L->offs[r];
            int64_t len = (int64_t)(L->offs[r + 1] - L->offs[r]);
            /* every position is entered, under its first kb bases with 'A' standing in for what lies beyond the
               reference's end or is not A/C/G/T: a lookup by a shorter prefix (a seed piece shorter than kb) then finds
               the positions near the end too; what the stand-ins let in by mistake fails the window check */
            for (int64_t p = 0; p < len; ++p) {
                uint64_t code = 0;
                for (int i = 0; i < kb; ++i) { int c = p + i < len ? code_of(s[p + i]) : 0; if (c < 0) c = 0; code = (code << 2) | (uint64_t)c; }
                if (pass == 0) { L->koff[code + 1]++; total++; }
                else { uint32_t at = L->koff[code]++; L->kref[at] = (uint32_t)r; L->kpos[at] = (uint32_t)p; }
            }
        }
        if (pass == 0) {
            for (uint64_t b = 0; b < nb; ++b) L->koff[b + 1] += L->koff[b];
            L->kref = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
            L->kpos = (uint32_t*)malloc((total ? total : 1) * sizeof(uint32_t));
        } else {
            /* koff[b] now holds end of bucket b: shift back */
            for (uint64_t b = nb; b > 0; --b) L->koff[b] = L->koff[b - 1];
            L->koff[0] = 0;
        }
    }
}

orc_lib* orc_lib_new(int32_t n, const char* seqs, const uint64_t* offs, int32_t kb) {
    orc_lib* L = (orc_lib*)calloc(1, sizeof(orc_lib));
    L->n = n;
    uint64_t tot = offs[n];
    L->text = (char*)malloc(tot + 1);
    L->offs = (uint64_t*)malloc((size_t)(n + 1) * sizeof(uint64_t));
    memcpy(L->offs, offs, (size_t)(n + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < tot; ++i) {
        char c = seqs[i];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        L->text[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';
    }
    L->text[tot] = 0;
    if (kb <= 0) kb = tot < (1u << 16) ? 6 : tot < (1u << 20) ? 8 : tot < (1u << 24) ? 10 : 12;
    L->kb = kb;
    lib_build_index(L);
    return L;
}
void orc_lib_free(orc_lib* L) {
    if (!L) return;
    free(L->text); free(L->offs); free(L->koff); free(L->kref); free(L->kpos); free(L);
}
int64_t orc_lib_total_len(const orc_lib* L) { return (int64_t)L->offs[L->n]; }

/* ------------------------------------------------------------- the policy */

typedef struct {
    char r[2][1024];    /* [1] = trimmed read ('+'), [0] = its reverse complement ('-') */
    int  l;             /* trimmed length */
    int  s;             /* seed length actually used: min(-l, l) */
    int  skip;          /* bowtie skips the read (empty after trimming, or l <= -n) */
} orc_query;

/* bowtie -5/-3 trim, then the seed is the first min(-l, len) bases of the READ
 * (its 5' end), whichever strand is being tried. */
static void query_init(orc_query* q, const char* read, int32_t len, const orc_round* P) {
    int l = len - P->trim5 - P->trim3;
    q->skip = 0;
    if (l <= 0 || l > 1000) { q->l = 0; q->s = 0; q->skip = 1; return; }     /* 1000 = MIRGE_MAX_READ_LEN: the library refuses longer reads */
    if (l <= P->seedmm) { q->l = l; q->s = 0; q->skip = 1; return; }  /* "length <= # seed mismatches" => skipped */
    q->l = l;
    q->s = P->seedlen < l ? P->seedlen : l;
    for (int i = 0; i < l; ++i) {
        char c = read[P->trim5 + i];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);          /* hypothesis H4: bowtie's read parser takes a/c/g/t as the base ... */
        q->r[1][i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N';   /* ... and any other character ('.', IUPAC codes) as N */
    }
    for (int i = 0; i < l; ++i) q->r[0][i] = comp_of(q->r[1][l - 1 - i]);
}

/* One window [off, off+l) of one reference on one strand.  Valid iff no
 * reference N, <= seedmm mismatches inside the seed, <= maxmm overall.  A read
 * N mismatches everything. */
static inline int window_ok(const char* ref, const orc_query* q, int fw, const orc_round* P, int* smm, int* tmm) {
    const char* r = q->r[fw];
    int l = q->l, s = q->s, sm = 0, tm = 0;
    int seed_lo = fw ? 0 : l - s, seed_hi = fw ? s : l;
    for (int i = 0; i < l; ++i) {
        char c = ref[i];
        if (c == 'N') return 0;
        if (r[i] == 'N' || r[i] != c) {
            tm++;
            if (i >= seed_lo && i < seed_hi) sm++;
            if (tm > P->maxmm) return 0;
        }
    }
    if (sm > P->seedmm) return 0;
    *smm = sm; *tmm = tm;
    return 1;
}

typedef struct { int stratum, mm, ref, off, fw; } cand;

/* H1 order inside a stratum: (mm, ref, off, fw) ascending; kept = maximum. */
static inline int cand_gt(const cand* a, const cand* b) {
    if (a->mm != b->mm) return a->mm > b->mm;
    if (a->ref != b->ref) return a->ref > b->ref;
    if (a->off != b->off) return a->off > b->off;
    return a->fw > b->fw;
}

typedef struct { int have; cand best; int minref, maxref; int64_t n; } acc;

static inline void acc_add(acc* A, const cand* c, int count) {
    if (!A->have || c->stratum < A->best.stratum) {
        A->have = 1; A->best = *c; A->minref = A->maxref = c->ref; A->n = count;
        return;
    }
    if (c->stratum > A->best.stratum) return;
    A->n += count;
    if (c->ref < A->minref) A->minref = c->ref;
    if (c->ref > A->maxref) A->maxref = c->ref;
    if (cand_gt(c, &A->best)) A->best = *c;
}

static void acc_out(const acc* A, orc_hit* out, int count_valid) {
    if (!A->have) { out->stratum = -1; out->ref = -1; out->off = -1; out->fw = -1; out->mm = -1; out->ambiguous = 0; out->nhits = count_valid ? 0 : -1; return; }
    out->stratum = A->best.stratum; out->ref = A->best.ref; out->off = A->best.off; out->fw = A->best.fw;
    out->mm = A->best.mm; out->ambiguous = A->minref != A->maxref; out->nhits = count_valid ? A->n : -1;
}

void orc_match_brute(const orc_lib* L, const char* read, int32_t len, const orc_round* P, orc_hit* out) {
    orc_query q; query_init(&q, read, len, P);
    acc A; memset(&A, 0, sizeof A);
    if (!q.skip) {
        for (int32_t r = 0; r < L->n; ++r) {
            const char* ref = L->text + L->offs[r];
            int64_t rl = (int64_t)(L->offs[r + 1] - L->offs[r]);
            for (int64_t o = 0; o + q.l <= rl; ++o)
                for (int fw = 1; fw >= 0; --fw) {
                    int sm, tm;
                    if (window_ok(ref + o, &q, fw, P, &sm, &tm)) {
                        cand c = { sm, tm, r, (int)o, fw };
                        acc_add(&A, &c, 1);
                    }
                }
        }
    }
    acc_out(&A, out, 1);
}

static int cand_cmp(const void* x, const void* y) {
    const cand* a = (const cand*)x; const cand* b = (const cand*)y;
    if (cand_gt(a, b)) return 1;
    if (cand_gt(b, a)) return -1;
    return 0;
}

int64_t orc_match_list(const orc_lib* L, const char* read, int32_t len, const orc_round* P, int32_t* hits, int64_t cap) {
    orc_query q; query_init(&q, read, len, P);
    if (q.skip) return 0;
    int64_t n = 0, alloc = 64; int best = 1 << 30;
    cand* v = (cand*)malloc((size_t)alloc * sizeof(cand));
    for (int32_t r = 0; r < L->n; ++r) {
        const char* ref = L->text + L->offs[r];
        int64_t rl = (int64_t)(L->offs[r + 1] - L->offs[r]);
        for (int64_t o = 0; o + q.l <= rl; ++o)
            for (int fw = 1; fw >= 0; --fw) {
                int sm, tm;
                if (!window_ok(ref + o, &q, fw, P, &sm, &tm)) continue;
                if (sm > best) continue;
                if (sm < best) { best = sm; n = 0; }
                if (n == alloc) { alloc *= 2; v = (cand*)realloc(v, (size_t)alloc * sizeof(cand)); }
                cand c = { sm, tm, r, (int)o, fw }; v[n++] = c;
            }
    }
    qsort(v, (size_t)n, sizeof(cand), cand_cmp);
    for (int64_t i = 0; i < n && i < cap; ++i) {
        hits[4 * i] = v[i].ref; hits[4 * i + 1] = v[i].off; hits[4 * i + 2] = v[i].fw; hits[4 * i + 3] = v[i].mm;
    }
    free(v);
    return n;
}

/* Indexed matcher.  Pigeonhole: the alignment has <= seedmm mismatches in the
 * seed and <= maxmm overall, so either (A) one of seedmm+1 pieces of the seed
 * or (B) one of maxmm+1 pieces of the whole read matches exactly; use whichever
 * gives the longer pieces.  Each piece is looked up by its first kb bases. */
void orc_match_indexed(const orc_lib* L, const char* read, int32_t len, const orc_round* P, orc_hit* out) {
    orc_query q; query_init(&q, read, len, P);
    acc A; memset(&A, 0, sizeof A);
    if (q.skip) { acc_out(&A, out, 0); return; }
    int nA = (P->seedmm < P->maxmm ? P->seedmm : P->maxmm) + 1, nB = P->maxmm + 1;
    int span = q.s, np = nA;
    if (q.l / nB > q.s / nA) { span = q.l; np = nB; }
    int kb = L->kb;
    if (span / np < 3) { orc_match_brute(L, read, len, P, out); out->nhits = -1; return; }
    /* candidate windows may repeat across pieces: max/min reductions do not care */
    for (int fw = 1; fw >= 0; --fw) {
        const char* r = q.r[fw];
        for (int p = 0; p < np; ++p) {
            int ps = (int)((int64_t)span * p / np), pe = (int)((int64_t)span * (p + 1) / np);   /* in READ coordinates */
            int qs = fw ? ps : q.l - pe;                       /* where that piece sits in the oriented read */
            int kq = pe - ps < kb ? pe - ps : kb;              /* a piece shorter than the index k-mer: every k-mer it is a prefix of */
            uint64_t code = 0; int ok = 1;
            for (int i = 0; i < kq; ++i) { int c = code_of(r[qs + i]); if (c < 0) { ok = 0; break; } code = (code << 2) | (uint64_t)c; }
            if (!ok) continue;                                  /* a piece holding a read N cannot be the exact one */
            uint64_t b0 = code << (2 * (kb - kq)), b1 = (code + 1) << (2 * (kb - kq));
            for (uint32_t at = L->koff[b0]; at < L->koff[b1]; ++at) {
                int32_t ref = (int32_t)L->kref[at];
                int64_t o = (int64_t)L->kpos[at] - qs;
                int64_t rl = (int64_t)(L->offs[ref + 1] - L->offs[ref]);
                if (o < 0 || o + q.l > rl) continue;
                int sm, tm;
                if (window_ok(L->text + L->offs[ref] + o, &q, fw, P, &sm, &tm)) {
                    cand c = { sm, tm, ref, (int)o, fw };
                    acc_add(&A, &c, 0);
                }
            }
        }
    }
    acc_out(&A, out, 0);
}

/* ---------------------------------------------------------------- collapse */

static inline uint64_t fnv1a(const char* s, uint64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    return h;
}

/* miRge.pl:300-307: key = the exact sequence line (chomp'ed). */
int64_t orc_collapse(int64_t n, const char* seqs, const uint64_t* offs, int64_t* uid, int64_t* rep) {
    uint64_t cap = 16; while (cap < (uint64_t)n * 2 + 16) cap <<= 1;
    int64_t* slot = (int64_t*)malloc(cap * sizeof(int64_t));
    for (uint64_t i = 0; i < cap; ++i) slot[i] = -1;
    int64_t U = 0;
    for (int64_t i = 0; i < n; ++i) {
        const char* s = seqs + offs[i]; uint64_t l = offs[i + 1] - offs[i];
        uint64_t h = fnv1a(s, l) & (cap - 1);
        for (;;) {
            int64_t u = slot[h];
            if (u < 0) { slot[h] = U; rep[U] = i; uid[i] = U; U++; break; }
            int64_t j = rep[u]; uint64_t lj = offs[j + 1] - offs[j];
            if (lj == l && memcmp(seqs + offs[j], s, l) == 0) { uid[i] = u; break; }
            h = (h + 1) & (cap - 1);
        }
    }
    free(slot);
    return U;
}

/* ----------------------------------------------------------------- cascade */

void orc_cascade(orc_lib* const* libs, int64_t U, const char* useqs, const uint64_t* uoffs,
                 int32_t nrounds, const orc_round* rounds, int32_t use_index, int32_t nthreads,
                 int8_t* ann_round, int32_t* ann_ref, int8_t* ann_amb, int64_t* stats) {
    for (int64_t u = 0; u < U; ++u) { ann_round[u] = 0; ann_ref[u] = -1; ann_amb[u] = 0; }
    (void)nthreads;
    for (int32_t r = 0; r < nrounds; ++r) {
        const orc_round* P = &rounds[r];
        const orc_lib* L = libs[P->lib];
        int64_t processed = 0, aligned = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads > 0 ? nthreads : 1) reduction(+:processed, aligned)
#endif
        for (int64_t u = 0; u < U; ++u) {
            if (ann_round[u] != 0) continue;                       /* miRge.pl:382 annot[0]==0 */
            int32_t len = (int32_t)(uoffs[u + 1] - uoffs[u]);
            if (P->len_lt && !(len < P->len_lt)) continue;         /* miRge.pl:383-386 */
            if (P->len_gt && !(len > P->len_gt)) continue;         /* miRge.pl:387-390 */
            processed++;
            orc_hit h;
            if (use_index) orc_match_indexed(L, useqs + uoffs[u], len, P, &h);
            else           orc_match_brute(L, useqs + uoffs[u], len, P, &h);
            if (h.ref >= 0) {                                      /* miRge.pl:445-446 */
                aligned++;
                ann_round[u] = (int8_t)(r + 1); ann_ref[u] = h.ref; ann_amb[u] = (int8_t)h.ambiguous;
            }
        }
        stats[2 * r] = processed; stats[2 * r + 1] = aligned;
    }
}

/* ---------------------------------------------------------- count vectors */

void orc_summarize(int64_t U, int32_t S, const uint32_t* counts, const int32_t* ulen,
                   const int8_t* ann_round, const int32_t* ann_ref, int32_t M,
                   uint64_t* quant, uint64_t* iscan, uint64_t* cat, uint64_t* tuniq,
                   int32_t maxlen, uint64_t* lenhist, int32_t n_rounds, const int32_t* round_lib) {
    /* round_lib == NULL: the reference's own five rounds, as miRge.pl:469-489 spells them out (annot 1 / 5 = miRNA,
     * 2 = hairpin, 3 = other, 4 = mRNA).  round_lib given (a cascade installed through mirge_set_rounds, which the
     * reference cannot express): the same rule stated by library -- a read's category is the library of the round that
     * annotated it, only rounds against the miRNA library index quant, only round 1 feeds iscan.  Both agree on the
     * reference cascade. */
    memset(quant, 0, (size_t)M * S * sizeof(uint64_t));
    memset(iscan, 0, (size_t)M * S * sizeof(uint64_t));
    memset(cat, 0, (size_t)5 * S * sizeof(uint64_t));
    memset(tuniq, 0, (size_t)S * sizeof(uint64_t));
    memset(lenhist, 0, (size_t)(maxlen + 1) * S * sizeof(uint64_t));
    for (int64_t u = 0; u < U; ++u)
        for (int32_t s = 0; s < S; ++s) {
            uint64_t c = counts[u * S + s];
            if (!c) continue;                                      /* miRge.pl:467 defined(quant[i]) */
            tuniq[s]++;                                            /* :468 */
            if (ulen[u] <= maxlen) lenhist[(int64_t)ulen[u] * S + s] += c;   /* :315 */
            int r = ann_round[u];
            if (round_lib) {
                if (r < 1 || r > n_rounds) { cat[4 * S + s] += c; continue; }
                int lib = round_lib[r - 1];
                cat[lib * S + s] += c;
                if (lib == 0) {
                    quant[(int64_t)ann_ref[u] * S + s] += c;
                    if (r == 1) iscan[(int64_t)ann_ref[u] * S + s] += c;
                }
                continue;
            }
            if (r == 1 || r == 5) {                                /* :469 */
                cat[0 * S + s] += c;                               /* :470 mirnaReads */
                quant[(int64_t)ann_ref[u] * S + s] += c;           /* :473 / :476 */
                if (r == 1) iscan[(int64_t)ann_ref[u] * S + s] += c;   /* :474 */
            } else if (r == 2) cat[1 * S + s] += c;                /* :480 hairpinReads */
            else if (r == 3)   cat[2 * S + s] += c;                /* :483 ornaReads */
            else if (r == 4)   cat[3 * S + s] += c;                /* :486 mrnaReads */
            else               cat[4 * S + s] += c;                /* :489 remReads */
        }
}
