/* trim_oracle.c -- CPU restatement of the read-trimming step that precedes the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, bench.py's baseline leg): the product never links or loads it.
 *
 * What the reference does (trim_file.py:26-68, miRge.pl:259-261, 268-289), per FASTQ record:
 *   1. QualityTrimmer(0, 10, base)        trim_file.py:32   BWA-style 3' quality trimming, cutoff 10
 *   2. AdapterCutter([3' adapters], e=.12) trim_file.py:41-43 or UnconditionalCutter(+N) (:36-37), or nothing ("none")
 *   3. keep if len >= 16                  trim_file.py:57
 *   4. `cutadapt -u trim5 -u -trim3`      miRge.pl:261      first/last bases removed unconditionally, no length filter
 * base = 64 iff the FIRST record holds a quality character > 'J' (the workers are started while the reader looks at
 * record 0, trim_file.py:123-131); the value printed to miRge.pl is decided over the first 1000 records.
 *
 * PARITY UNPINNED: steps 1, 2 and 4 are cutadapt's (third party, ">= 1.8.1, tested 1.8.1-1.11", README.md:23-29,
 * absent from /root/reference and not installed); the reference holds no test vector for them.  This file restates
 * cutadapt's published algorithms as of the 1.9-1.11 sources: quality_trim_index (qualtrim), Adapter.match_to for a
 * 3' ("back") adapter = leftmost exact occurrence first, else Aligner.locate (semiglobal edit-distance DP with
 * Ukkonen's band, flags START_WITHIN_SEQ2 | STOP_WITHIN_SEQ2 | STOP_WITHIN_SEQ1, best = most matches, then lowest cost,
 * error bound cost <= aligned_adapter_length * rate evaluated in double, min overlap 3), AdapterCutter (times = 1, best
 * adapter = most matches, first wins ties), UnconditionalCutter.  Adapters with IUPAC wildcard characters: see
 * chars_equal below (no exact-occurrence shortcut for them; error bound on the full aligned length, as in 1.9-1.11 --
 * later versions discount N positions).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <ctype.h>

#define ORC_TRIM_MAXADAPTER 64

/* cutadapt qualtrim.quality_trim_index(qualities, cutoff_front, cutoff_back, base) */
void orc_quality_trim(const char* q, int n, int cutoff_front, int cutoff_back, int base, int* start_out, int* stop_out) {
    int s = 0, max_qual = 0, start = 0, stop = n;
    for (int i = 0; i < n; ++i) {
        s += cutoff_front - ((int)(unsigned char)q[i] - base);
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; start = i + 1; }
    }
    max_qual = 0; s = 0;
    for (int i = n - 1; i >= 0; --i) {
        s += cutoff_back - ((int)(unsigned char)q[i] - base);
        if (s < 0) break;
        if (s > max_qual) { max_qual = s; stop = i; }
    }
    if (start >= stop) { start = 0; stop = 0; }
    *start_out = start; *stop_out = stop;
}

typedef struct { int cost, matches, origin; } Entry;

/* Adapter wildcards (cutadapt Adapter(..., adapter_wildcards=True), the default: active when the adapter holds a character
 * outside ACGT).  align.pyx: the adapter is translated with the IUPAC table (a bit per base it stands for), the read with
 * the ACGT table (every other character 0), and two characters are equal when their codes share a bit -- so an adapter N
 * matches any read base, a read N (or any other non-base) matches nothing, X matches nothing. */
static int iupac_bits(int ch) {
    switch (ch) {
        case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
        case 'R': return 1 | 4; case 'Y': return 2 | 8; case 'S': return 2 | 4; case 'W': return 1 | 8;
        case 'K': return 4 | 8; case 'M': return 1 | 2; case 'B': return 2 | 4 | 8; case 'D': return 1 | 4 | 8;
        case 'H': return 1 | 2 | 8; case 'V': return 1 | 2 | 4; case 'N': return 15; case 'X': return 0;
        default: return -1;
    }
}
static int acgt_bits(int ch) { return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 4 : ch == 'T' ? 8 : 0; }
static int chars_equal(int adapter_ch, int read_ch, int wildcards) {
    return wildcards ? (iupac_bits(adapter_ch) & acgt_bits(read_ch)) != 0 : adapter_ch == read_ch;
}
/* 1 = the adapter holds wildcards, 0 = plain ACGT, -1 = a character cutadapt's parser refuses */
int orc_adapter_wildcards(const char* adapter, int m) {
    int wild = 0;
    for (int i = 0; i < m; ++i) {
        if (iupac_bits((unsigned char)adapter[i]) < 0) return -1;
        if (!strchr("ACGT", adapter[i])) wild = 1;
    }
    return wild;
}

/* cutadapt _align.Aligner.locate for a back adapter (reference = adapter, query = read).
 * out = {astart, astop, rstart, rstop, matches, errors}; returns 1 if an alignment qualifies. */
/* anchored (a 3' adapter written "ADAPTER$": where = SUFFIX, flags = START_WITHIN_SEQ2 alone): the alignment must take in
 * the whole adapter and end at the read's end -- only cell (m, n) is a candidate. */
static int locate_back(const char* s1, int m, const char* s2, int n, double max_error_rate, int min_overlap, int anchored, int wildcards, int out[6]) {
    Entry column[ORC_TRIM_MAXADAPTER + 1];
    const int k = (int)(max_error_rate * m);
    for (int i = 0; i <= m; ++i) { column[i].matches = 0; column[i].cost = i; column[i].origin = 0; }
    struct { int ref_stop, query_stop, cost, origin, matches; } best = {m, n, m + n, 0, 0};
    int last = m < k + 1 ? m : k + 1;
    for (int j = 1; j <= n; ++j) {
        Entry tmp = column[0];
        column[0].origin = j;                                    /* the alignment may start anywhere in the read */
        for (int i = 1; i <= last; ++i) {
            int cost, origin, matches;
            if (chars_equal((unsigned char)s1[i - 1], (unsigned char)s2[j - 1], wildcards)) {
                cost = tmp.cost; origin = tmp.origin; matches = tmp.matches + 1;
            } else {
                int cost_diag = tmp.cost + 1, cost_deletion = column[i].cost + 1, cost_insertion = column[i - 1].cost + 1;
                if (cost_diag <= cost_deletion && cost_diag <= cost_insertion) { cost = cost_diag; origin = tmp.origin; matches = tmp.matches; }
                else if (cost_insertion <= cost_deletion) { cost = cost_insertion; origin = column[i - 1].origin; matches = column[i - 1].matches; }
                else { cost = cost_deletion; origin = column[i].origin; matches = column[i].matches; }
            }
            tmp = column[i];
            column[i].cost = cost; column[i].origin = origin; column[i].matches = matches;
        }
        while (last >= 0 && column[last].cost > k) --last;
        if (last < m) ++last;
        else if (!anchored) {
            /* the whole adapter is aligned and ends inside the read */
            int length = m + (column[m].origin < 0 ? column[m].origin : 0);
            int cost = column[m].cost, matches = column[m].matches;
            if (length >= min_overlap && (double)cost <= length * max_error_rate &&
                (matches > best.matches || (matches == best.matches && cost < best.cost))) {
                best.matches = matches; best.cost = cost; best.origin = column[m].origin; best.ref_stop = m; best.query_stop = j;
                if (cost == 0 && matches == m) break;
            }
        }
    }
    /* the read may end inside the adapter: last column, every adapter prefix.  (After the early exit above nothing
       here can win: it would need more than m matches.  Rows the band no longer reaches hold a cost > k from the
       column in which they were dropped, so they cannot qualify either.) */
    for (int i = m; i >= (anchored ? m : 0); --i) {
        int length = i + (column[i].origin < 0 ? column[i].origin : 0);
        int cost = column[i].cost, matches = column[i].matches;
        if (length >= min_overlap && (double)cost <= length * max_error_rate &&
            (matches > best.matches || (matches == best.matches && cost < best.cost))) {
            best.matches = matches; best.cost = cost; best.origin = column[i].origin; best.ref_stop = i; best.query_stop = n;
        }
    }
    if (best.cost == m + n) return 0;
    int start1 = best.origin >= 0 ? 0 : -best.origin, start2 = best.origin >= 0 ? best.origin : 0;
    out[0] = start1; out[1] = best.ref_stop; out[2] = start2; out[3] = best.query_stop; out[4] = best.matches; out[5] = best.cost;
    return 1;
}

/* cutadapt Adapter(where=BACK or SUFFIX, max_error_rate, min_overlap=3, indels=True).match_to(read): the read is
 * compared upper-cased (read_seq = read.sequence.upper()), the adapter was upper-cased with U -> T when it was made. */
int orc_adapter_back(const char* adapter, int m, const char* read_in, int n, double max_error_rate, int min_overlap, int anchored, int out[6]) {
    if (m <= 0 || m > ORC_TRIM_MAXADAPTER || n < 0 || n > 4096) return 0;
    char read[4097];
    for (int i = 0; i < n; ++i) read[i] = (char)toupper((unsigned char)read_in[i]);
    const int wildcards = orc_adapter_wildcards(adapter, m);
    if (wildcards < 0) return 0;
    if (wildcards) {
        /* match_to: "try to find an exact match first unless wildcards are allowed" -- straight to the aligner */
    } else if (anchored) {
        if (n >= m && memcmp(read + n - m, adapter, (size_t)m) == 0) { out[0] = 0; out[1] = m; out[2] = n - m; out[3] = n; out[4] = m; out[5] = 0; return 1; }
    } else {
        for (int p = 0; p + m <= n; ++p)
            if (memcmp(read + p, adapter, (size_t)m) == 0) { out[0] = 0; out[1] = m; out[2] = p; out[3] = p + m; out[4] = m; out[5] = 0; return 1; }
    }
    if (!locate_back(adapter, m, read, n, max_error_rate, min_overlap, anchored, wildcards, out)) return 0;
    int length = out[1] - out[0];
    if (length < min_overlap || (double)out[5] / length > max_error_rate) return 0;
    return 1;
}

/* One record through trim_file.py's Worker and the `cutadapt -u` pass.  adapters: "none", "+N", or comma-separated
 * 3' adapter sequences.  Returns 1 if the record is kept; [*out_start, *out_start + *out_len) = what is left of seq. */
int orc_trim_record(const char* seq, const char* qual, int n, const char* adapters, int base, int trim5, int trim3,
                    int* out_start, int* out_len) {
    int a, b;
    orc_quality_trim(qual, n, 0, 10, base, &a, &b);
    if (adapters[0] == '+' || adapters[0] == '-') {
        int u = atoi(adapters);                                   /* UnconditionalCutter(int(adapter)) */
        if (u > 0) a = a + u < b ? a + u : b;
        else if (u < 0) b = b + u > a ? b + u : a;
    } else if (strcmp(adapters, "none") != 0) {
        /* AdapterCutter: best match over the adapters (most matches; the first wins ties), times = 1 */
        int best_matches = -1, best_rstart = -1;
        const char* p = adapters;
        while (*p) {
            /* gather_adapters (trim_file.py:17): "name=SEQ", a trailing "$" anchors the adapter at the read's 3' end;
               Adapter.__init__: upper case, U -> T */
            const char* e = strchr(p, ','); int m = e ? (int)(e - p) : (int)strlen(p);
            const char* q = p; int ql = m;
            const char* eq = memchr(q, '=', (size_t)ql);
            if (eq) { ql -= (int)(eq + 1 - q); q = eq + 1; }
            int anchored = ql > 0 && q[ql - 1] == '$';
            if (anchored) --ql;
            char ad[ORC_TRIM_MAXADAPTER + 1];
            int out[6];
            if (ql > 0 && ql <= ORC_TRIM_MAXADAPTER) {
                for (int i = 0; i < ql; ++i) { char c = (char)toupper((unsigned char)q[i]); ad[i] = c == 'U' ? 'T' : c; }
                if (orc_adapter_back(ad, ql, seq + a, b - a, 0.12, 3, anchored, out) && out[4] > best_matches) { best_matches = out[4]; best_rstart = out[2]; }
            }
            p += m; if (*p == ',') ++p;
        }
        if (best_rstart >= 0) b = a + best_rstart;
    }
    if (b - a < 16) return 0;                                     /* trim_file.py:57 */
    if (trim5 > 0) a = a + trim5 < b ? a + trim5 : b;             /* cutadapt -u trim5 */
    if (trim3 > 0) b = b - trim3 > a ? b - trim3 : a;             /* cutadapt -u -trim3 */
    *out_start = a; *out_len = b - a;
    return 1;
}

/* A whole FASTQ text (four lines per record, as cutadapt's FastqReader and miRge.pl:299-307 agree on for well-formed
 * files).  keep/start/len per record (start relative to the record's sequence line); returns the number of records,
 * or -1 if cap is too small, -2 if a record's sequence and quality lines differ in length.
 * phred[0] = base the workers used (decided on record 0), phred[1] = value reported to miRge.pl (first 1000 records). */
int64_t orc_trim_fastq(const char* data, uint64_t size, const char* adapters, int trim5, int trim3, int64_t cap,
                       int32_t* keep, int32_t* start, int32_t* len, uint64_t* seq_off, int32_t* phred) {
    int64_t nrec = 0; uint64_t i = 0;
    int base = 33, reported = 33;
    /* pass 1: phred detection */
    {
        uint64_t p = 0; int64_t r = 0;
        while (p < size && r < 1000) {
            uint64_t ls[4], le[4];
            for (int k = 0; k < 4; ++k) { ls[k] = p; while (p < size && data[p] != '\n') ++p; le[k] = p; if (p < size) ++p; }
            while (le[3] > ls[3] && data[le[3] - 1] == '\r') --le[3];
            int hi = 0;
            for (uint64_t x = ls[3]; x < le[3]; ++x) if ((unsigned char)data[x] > 74) hi = 1;
            if (hi) { reported = 64; if (r == 0) base = 64; }
            ++r;
        }
    }
    phred[0] = base; phred[1] = reported;
    while (i < size) {
        uint64_t ls[4], le[4];
        for (int k = 0; k < 4; ++k) { ls[k] = i; while (i < size && data[i] != '\n') ++i; le[k] = i; if (i < size) ++i; }
        if (nrec >= cap) return -1;
        for (int k = 1; k < 4; k += 2) while (le[k] > ls[k] && data[le[k] - 1] == '\r') --le[k];   /* FastqReader strips line ends */
        int n = (int)(le[1] - ls[1]);
        if ((int)(le[3] - ls[3]) != n) return -2;
        int a = 0, l = 0;
        keep[nrec] = orc_trim_record(data + ls[1], data + ls[3], n, adapters, base, trim5, trim3, &a, &l);
        start[nrec] = a; len[nrec] = l; seq_off[nrec] = ls[1];
        ++nrec;
    }
    return nrec;
}
