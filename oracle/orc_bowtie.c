/*
 * orc_bowtie.c -- bowtie-1-CLI-compatible front end over the CPU oracle.
 * TEST INFRASTRUCTURE ONLY: used to run the unmodified reference miRge.pl in
 * the build container (golden-fixture generation, tests/golden/gen/) where no
 * real bowtie exists.  Contract restated from miRge.pl:327-339, 353 (command
 * line), :409-413 (stderr summary), :428-430 (TSV columns 0,1,2) and :457-459
 * (`<exe>-inspect -n <base>` prints reference names).
 *
 *   orc-bowtie -a --best --strata --threads N [--phred64-quals] <base> -n K
 *              [-l L] [-5 A] [-3 B] [-e E] -f <reads.fa>
 *   orc-bowtie-inspect -n <base>
 *
 * <base> is the bowtie index basename; the sequences are read from <base>.fa.
 * Every best-stratum alignment is printed (as `-a --best --strata` does), in
 * hypothesis-H1 order, so the line miRge keeps is the oracle's representative.
 */
#include "mirge_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char** names; char* seqs; uint64_t* offs; int32_t n; } fasta;

static void read_fasta(const char* path, fasta* F) {
    FILE* f = fopen(path, "r");
    if (!f) { fprintf(stderr, "orc-bowtie: cannot open %s\n", path); exit(1); }
    size_t cap = 1 << 16, len = 0; int32_t ncap = 64;
    F->seqs = (char*)malloc(cap); F->names = (char**)malloc(ncap * sizeof(char*));
    F->offs = (uint64_t*)malloc((ncap + 1) * sizeof(uint64_t)); F->n = 0;
    char* line = NULL; size_t lcap = 0; ssize_t got;
    while ((got = getline(&line, &lcap, f)) >= 0) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (line[0] == '>') {
            if (F->n == ncap) { ncap *= 2; F->names = (char**)realloc(F->names, ncap * sizeof(char*)); F->offs = (uint64_t*)realloc(F->offs, (ncap + 1) * sizeof(uint64_t)); }
            F->names[F->n] = strdup(line + 1); F->offs[F->n] = len; F->n++;
        } else if (F->n > 0) {
            if (len + (size_t)got + 1 > cap) { while (len + (size_t)got + 1 > cap) cap *= 2; F->seqs = (char*)realloc(F->seqs, cap); }
            memcpy(F->seqs + len, line, (size_t)got); len += (size_t)got;
        }
    }
    F->offs[F->n] = len;
    free(line); fclose(f);
}

int main(int argc, char** argv) {
    const char* exe = strrchr(argv[0], '/'); exe = exe ? exe + 1 : argv[0];
    int inspect = strstr(exe, "-inspect") != NULL;
    orc_round P = { 0, 0, 0, 0, 0, 28, 2, 2 };   /* bowtie defaults: -l 28 -n 2 -e 70 */
    const char* pos[2] = { NULL, NULL }; int npos = 0, e = 70;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (!strcmp(a, "-n")) { if (inspect) continue; P.seedmm = atoi(argv[++i]); }
        else if (!strcmp(a, "-l")) P.seedlen = atoi(argv[++i]);
        else if (!strcmp(a, "-5")) P.trim5 = atoi(argv[++i]);
        else if (!strcmp(a, "-3")) P.trim3 = atoi(argv[++i]);
        else if (!strcmp(a, "-e")) e = atoi(argv[++i]);
        else if (!strcmp(a, "--threads") || !strcmp(a, "-p")) ++i;
        else if (a[0] == '-') continue;                    /* -a --best --strata -f --phred64-quals: flags */
        else if (npos < 2) pos[npos++] = a;
    }
    P.maxmm = e / 30;
    if (!pos[0]) { fprintf(stderr, "orc-bowtie: no index given\n"); return 1; }
    char path[4096]; snprintf(path, sizeof path, "%s.fa", pos[0]);
    fasta F; read_fasta(path, &F);
    if (inspect) { for (int32_t i = 0; i < F.n; ++i) printf("%s\n", F.names[i]); return 0; }
    if (!pos[1]) { fprintf(stderr, "orc-bowtie: no reads given\n"); return 1; }
    orc_lib* L = orc_lib_new(F.n, F.seqs, F.offs, 0);
    fasta R; read_fasta(pos[1], &R);
    int64_t processed = 0, aligned = 0, cap = 1 << 16;
    int32_t* hits = (int32_t*)malloc((size_t)cap * 4 * sizeof(int32_t));
    for (int32_t i = 0; i < R.n; ++i) {
        const char* s = R.seqs + R.offs[i]; int32_t len = (int32_t)(R.offs[i + 1] - R.offs[i]);
        processed++;
        int64_t n = orc_match_list(L, s, len, &P, hits, cap);
        if (n > cap) n = cap;
        if (n > 0) aligned++;
        for (int64_t h = 0; h < n; ++h) {
            char nm[1024]; strncpy(nm, F.names[hits[4 * h]], sizeof nm - 1); nm[sizeof nm - 1] = 0;
            char* sp = strpbrk(nm, " \t"); if (sp) *sp = 0;      /* bowtie prints the name up to the first whitespace */
            printf("%s\t%c\t%s\t%d\t%.*s\t", R.names[i], hits[4 * h + 2] ? '+' : '-', nm, hits[4 * h + 1], len, s);
            for (int32_t k = 0; k < len; ++k) putchar('I');
            printf("\t%lld\t\n", (long long)(n - 1));
        }
    }
    fprintf(stderr, "# reads processed: %lld\n# reads with at least one reported alignment: %lld (%.2f%%)\n"
                    "# reads that failed to align: %lld (%.2f%%)\nReported %lld alignments to 1 output stream(s)\n",
            (long long)processed, (long long)aligned, processed ? 100.0 * aligned / processed : 0.0,
            (long long)(processed - aligned), processed ? 100.0 * (processed - aligned) / processed : 0.0, (long long)aligned);
    return 0;
}
