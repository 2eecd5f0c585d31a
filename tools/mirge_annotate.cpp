// mirge-annotate -- batch mode without Perl over libmirge_hip.so: what miRge.pl:133-150 does between the
// command line and the tables, minus graphs / HTML / isomiR tables:
//   per sample  trimRaw + `cutadapt -u` + quantReads (miRge.pl:259-262)   -> mirge_collapse_raw_fastq   (--adapter given)
//               or quantReads on an already trimmed file (:291-319)        -> mirge_collapse_fastq
//   runAnnotationPipeline (:321-372) + summarize (:453-494)               -> mirge_annotate
//   miRNAmerge, filter, writeDataToCSV (:496-554, 690-907)                -> mirge_write_tables
// Option names follow miRge.pl:59 where they exist (--species is a directory here: miRge.seqLibs/<species>).
//   mirge-annotate --species DIR --SampleFiles a.fastq,b.fastq.gz [--SampleNames a,b] [--adapter illumina|ion|none|+N|SEQ]
//                  [--trim5 N] [--trim3 N] [--outputDir DIR] [--device N | --gpus N [--host-staged]]
// --gpus N = one context per GPU inside the library (mirge_group_*, RCCL); sample files may be .gz and are read / inflated
// ahead of the device(s) by the host's threads (mirge_collapse_fastq_batch).
#include "../include/mirge_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::vector<std::string> split(const std::string& s) {
    std::vector<std::string> v; size_t a = 0;
    for (;;) { size_t b = s.find(',', a); v.emplace_back(s, a, b == std::string::npos ? std::string::npos : b - a); if (b == std::string::npos) break; a = b + 1; }
    return v;
}

int main(int argc, char** argv) {
    std::string species, files, names, adapter, out = "miRge.out"; int trim5 = 0, trim3 = 0, device = 0, gpus = 0; bool raw = false, staged = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], v; size_t eq = a.find('=');
        while (a.size() > 1 && a[0] == '-') a.erase(0, 1);
        eq = a.find('=');
        if (a == "host-staged") { staged = true; continue; }
        if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); } else if (i + 1 < argc) v = argv[++i];
        if (a == "species") species = v; else if (a == "SampleFiles") files = v; else if (a == "SampleNames") names = v;
        else if (a == "adapter") { adapter = v; raw = true; } else if (a == "trim5") trim5 = atoi(v.c_str()); else if (a == "trim3") trim3 = atoi(v.c_str());
        else if (a == "outputDir") out = v; else if (a == "device") device = atoi(v.c_str()); else if (a == "gpus") gpus = atoi(v.c_str());
        else { fprintf(stderr, "mirge-annotate: unknown option --%s\n", a.c_str()); return 2; }
    }
    if (species.empty() || files.empty()) { fprintf(stderr, "usage: mirge-annotate --species DIR --SampleFiles a.fastq,b.fastq.gz [--SampleNames a,b] [--adapter X] [--trim5 N] [--trim3 N] [--outputDir DIR] [--device N | --gpus N [--host-staged]]\n"); return 2; }
    std::vector<std::string> F = split(files), N = names.empty() ? F : split(names);
    if (N.size() != F.size()) { fprintf(stderr, "mirge-annotate: %zu sample names for %zu sample files\n", N.size(), F.size()); return 2; }
    const int32_t S = (int32_t)F.size();
    // --gpus N: one context per GPU inside the library (mirge_group_*: whole sample files per GPU, unique tables merged by key
    // owner over RCCL, one all-reduce of the count vectors).  --host-staged: the N ranks share --device and exchange through
    // host memory (rehearsal on a one-GPU machine).  Without --gpus: one context on --device.
    mirge_group* g = nullptr; mirge_ctx* c = nullptr;
    if (gpus > 0) {
        std::vector<int32_t> dev((size_t)gpus);
        for (int r = 0; r < gpus; ++r) dev[(size_t)r] = staged ? device : r;
        g = mirge_group_open_ex(species.c_str(), S, gpus, dev.data(), staged ? MIRGE_GROUP_HOST_STAGED : 0u);
        if (!g) { fprintf(stderr, "mirge-annotate: %s\n", mirge_group_last_error(nullptr)); return 1; }
    } else {
        c = mirge_open(species.c_str(), S, device);
        if (!c) { fprintf(stderr, "mirge-annotate: %s\n", mirge_last_error(nullptr)); return 1; }
    }
    auto die = [&](const std::string& what) {
        fprintf(stderr, "mirge-annotate: %s: %s\n", what.c_str(), g ? mirge_group_last_error(g) : mirge_last_error(c));
        if (g) mirge_group_close(g); else mirge_close(c);
        return 1;
    };
    {   // all sample files in one call: host threads read / inflate files ahead of the device(s)
        std::vector<const char*> fp; std::vector<int32_t> smp((size_t)S);
        for (int32_t i = 0; i < S; ++i) { fp.push_back(F[(size_t)i].c_str()); smp[(size_t)i] = i; }
        std::vector<uint64_t> total((size_t)S), kept((size_t)S); std::vector<int32_t> phred((size_t)S);
        const int rc = g ? mirge_group_collapse_fastq_batch(g, S, fp.data(), smp.data(), raw ? adapter.c_str() : nullptr, trim5, trim3, total.data(), kept.data(), phred.data())
                         : mirge_collapse_fastq_batch(c, S, fp.data(), smp.data(), raw ? adapter.c_str() : nullptr, trim5, trim3, total.data(), kept.data(), phred.data());
        if (rc) return die("sample files");
        for (int32_t i = 0; i < S; ++i) printf("%s: totalReads %llu trimmedReads %llu\n", N[(size_t)i].c_str(), (unsigned long long)total[(size_t)i], (unsigned long long)kept[(size_t)i]);
    }
    if (g ? mirge_group_annotate(g) : mirge_annotate(c)) return die("annotation");
    static const char* rn[5] = {"exact miRNA", "hairpin miRNA", "non miRNA/mRNA RNA", "mRNA", "isomiR miRNA"};      // miRge.pl:111
    for (int r = 1; r <= 5; ++r) {
        uint64_t p = 0, a = 0; double sec = 0;
        if (g ? mirge_group_round_stats(g, r, &p, &a, &sec) : mirge_round_stats(c, r, &p, &a, &sec)) return die("round statistics");
        printf("%s: readsProcessed %llu readsAligned %llu cpuTime %.6f\n", rn[r - 1], (unsigned long long)p, (unsigned long long)a, sec);
    }
    if (g) { double m = 0, ar = 0; (void)mirge_group_timing(g, &m, &ar); printf("%d ranks: merge of the unique tables %.3f ms, count all-reduce %.3f ms\n", gpus, m, ar); }
    {   // how much of the result rides on hypothesis H1 (which of several equally good references bowtie prints last):
        // uniques / reads whose best stratum held more than one reference (parity with bowtie unpinned)
        uint64_t ua = 0, ut = 0, ra = 0, rt = 0;
        for (int r = 0; r < (g ? gpus : 1); ++r) {
            mirge_ctx* x = g ? mirge_group_ctx(g, r) : c;
            const uint64_t U = mirge_n_unique(x), E = mirge_n_counts(x);
            std::vector<int8_t> rnd(U ? U : 1); std::vector<uint8_t> amb(U ? U : 1);
            std::vector<uint64_t> uid(E ? E : 1); std::vector<uint32_t> cnt(E ? E : 1);
            if (mirge_export_uniques(x, nullptr, 0, nullptr, rnd.data(), nullptr, amb.data()) || mirge_export_counts(x, uid.data(), nullptr, cnt.data())) {
                fprintf(stderr, "mirge-annotate: per-unique results: %s\n", mirge_last_error(x)); if (g) mirge_group_close(g); else mirge_close(c); return 1;
            }
            for (uint64_t u = 0; u < U; ++u) if (rnd[u] > 0) { ++ua; ut += amb[u] != 0; }
            for (uint64_t e = 0; e < E; ++e) if (rnd[uid[e]] > 0) { ra += cnt[e]; rt += amb[uid[e]] ? cnt[e] : 0; }
        }
        printf("annotated: %llu uniques / %llu reads; best stratum held several references (assignment follows hypothesis H1): %llu uniques (%.2f%%) / %llu reads (%.2f%%)\n",
               (unsigned long long)ua, (unsigned long long)ra, (unsigned long long)ut, ua ? 100.0 * ut / ua : 0.0, (unsigned long long)rt, ra ? 100.0 * rt / ra : 0.0);
    }
    std::vector<const char*> np; for (auto& s : N) np.push_back(s.c_str());
    std::vector<uint64_t> tot(F.size()), uq(F.size());
    const std::string merges = species + "/merges.csv";
    if (g ? mirge_group_write_tables(g, merges.c_str(), out.c_str(), np.data(), tot.data(), uq.data())
          : mirge_write_tables(c, merges.c_str(), out.c_str(), np.data(), tot.data(), uq.data())) return die("tables");
    for (size_t i = 0; i < F.size(); ++i) printf("%s: mirnaReadsFiltered %llu mirnaUniqFiltered %llu\n", N[i].c_str(), (unsigned long long)tot[i], (unsigned long long)uq[i]);
    if (g) mirge_group_close(g); else mirge_close(c);
    return 0;
}
