// mirge-bowtie -- bowtie-1-CLI-compatible front end over libmirge_hip.so.
//
// The reference's own substitution point is the bowtie executable (`--bowtie <exe>`, miRge.pl:59, 91):
// with this program there, the UNMODIFIED miRge.pl runs its five annotation rounds on the GPU.
// Contract restated from the reference (SURVEY.md 8b "B1"):
//   miRge.pl:327-339, 353   <exe> -a --best --strata --threads N [--phred64-quals] <ebwt_base> -n K
//                           [-l L] [-5 A] [-3 B] -f <reads.fa>  1><hits> 2><log>   (exit status 0 required)
//   miRge.pl:428-430        stdout: TAB-separated, col 0 = read name, col 1 = strand, col 2 = reference name
//                           (up to the first whitespace); no line for an unaligned read; the LAST line of
//                           a read wins
//   miRge.pl:409-413        stderr: "reads processed: <n>" and "reads with at least one reported alignment: <n>"
//   miRge.pl:457-459        <exe>-inspect -n <ebwt_base>: one reference name (whole header) per line
// The sequences are read from <ebwt_base>.fa beside the .ebwt files, else out of <ebwt_base>.{1,3,4}.ebwt.  One line is printed per aligned read:
// the representative of its best stratum (the line miRge would keep); columns other than 0 and 2 are
// placeholders (miRge ignores them; column 1 is stored and never used).
//
// This is the slow way in: one process start, one library load and one index build per round, as with
// bowtie itself.  The fast way is the in-process C ABI (include/mirge_hip.h, perl/MirgeHip.xs).
#include "../mirge_amd/csrc/ebwt.h"
#include "../include/mirge_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

struct Fasta { std::vector<std::string> names; std::string seqs; std::vector<uint64_t> offs; };

static bool read_fasta(const std::string& path, Fasta& F) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string data; char buf[1 << 16]; size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.append(buf, got);
    fclose(f);
    size_t i = 0, n = data.size();
    while (i < n) {
        size_t e = data.find('\n', i); if (e == std::string::npos) e = n;
        size_t le = e; while (le > i && data[le - 1] == '\r') --le;
        if (le > i && data[i] == '>') { F.names.emplace_back(data, i + 1, le - i - 1); F.offs.push_back(F.seqs.size()); }
        else if (!F.names.empty()) F.seqs.append(data, i, le - i);
        i = e + 1;
    }
    F.offs.push_back(F.seqs.size());
    return true;
}

int main(int argc, char** argv) {
    const char* exe = strrchr(argv[0], '/'); exe = exe ? exe + 1 : argv[0];
    const bool inspect = strstr(exe, "-inspect") != nullptr;
    mirge_round R = {0, 0, 0, 0, 0, 28, 2, 2};          // bowtie defaults: -l 28 -n 2 -e 70
    int e = 70; const char* pos[2] = {nullptr, nullptr}; int npos = 0;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        auto val = [&](int& dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (!strcmp(a, "-n")) { if (!inspect) val(R.seedmm); }
        else if (!strcmp(a, "-l")) val(R.seedlen);
        else if (!strcmp(a, "-5")) val(R.trim5);
        else if (!strcmp(a, "-3")) val(R.trim3);
        else if (!strcmp(a, "-e")) val(e);
        else if (!strcmp(a, "--threads") || !strcmp(a, "-p")) ++i;
        else if (a[0] == '-') continue;                  // -a --best --strata -f --phred64-quals: flags
        else if (npos < 2) pos[npos++] = a;
    }
    R.maxmm = e / 30; if (R.maxmm > 3) R.maxmm = 3;
    if (!pos[0]) { fprintf(stderr, "mirge-bowtie: no index basename given\n"); return 1; }
    Fasta lib;
    if (!read_fasta(std::string(pos[0]) + ".fa", lib) && !read_fasta(std::string(pos[0]) + ".fasta", lib)) {
        // no FASTA beside the index: the sequences come out of the bowtie index itself (ebwt.h; layout unvalidated)
        std::string err;
        if (mg::ebwt_extract(pos[0], lib.names, lib.seqs, lib.offs, err) != 0) {
            fprintf(stderr, "mirge-bowtie: cannot read %s.fa, and: %s\n", pos[0], err.c_str());
            return 1;
        }
    }
    if (inspect) { for (auto& nm : lib.names) printf("%s\n", nm.c_str()); return 0; }
    if (!pos[1]) { fprintf(stderr, "mirge-bowtie: no reads file given\n"); return 1; }
    Fasta reads;
    if (!read_fasta(pos[1], reads)) { fprintf(stderr, "mirge-bowtie: cannot read %s\n", pos[1]); return 1; }

    mirge_ctx* c = mirge_open(nullptr, 1, 0);
    if (!c) { fprintf(stderr, "mirge-bowtie: %s\n", mirge_last_error(nullptr)); return 1; }
    auto die = [&](const char* what) { fprintf(stderr, "mirge-bowtie: %s: %s\n", what, mirge_last_error(c)); mirge_close(c); return 1; };
    std::string blob; for (auto& nm : lib.names) { blob += nm; blob.push_back('\0'); }
    if (mirge_load_library_mem(c, 0, (int32_t)lib.names.size(), blob.data(), lib.seqs.data(), lib.offs.data())) return die("library");
    if (mirge_set_rounds(c, 1, &R)) return die("options");
    if (mirge_collapse_seqs(c, reads.seqs.data(), reads.offs.data(), reads.names.size(), 0)) return die("reads");
    if (mirge_annotate(c)) return die("align");
    uint64_t U = mirge_n_unique(c);
    std::unordered_map<std::string, int32_t> hit;
    hit.reserve((size_t)U * 2);
    for (uint64_t u = 0; u < U; ++u) {
        const char* seq; int32_t ref[MIRGE_MAX_ROUNDS]; uint8_t amb; const uint32_t* cnt;
        if (mirge_unique(c, u, &seq, ref, &amb, &cnt)) return die("results");
        if (ref[0] >= 0) hit.emplace(seq, ref[0]);
    }
    uint64_t aligned = 0;
    for (size_t i = 0; i < reads.names.size(); ++i) {
        std::string s(reads.seqs, reads.offs[i], reads.offs[i + 1] - reads.offs[i]);
        auto it = hit.find(s);
        if (it == hit.end()) continue;
        ++aligned;
        printf("%s\t+\t%s\t0\t%s\t%s\t0\t\n", reads.names[i].c_str(), mirge_lib_short_name(c, 0, it->second), s.c_str(),
               std::string(s.size(), 'I').c_str());
    }
    uint64_t n = reads.names.size();
    fprintf(stderr, "# reads processed: %llu\n# reads with at least one reported alignment: %llu (%.2f%%)\n"
                    "# reads that failed to align: %llu (%.2f%%)\nReported %llu alignments to 1 output stream(s)\n",
            (unsigned long long)n, (unsigned long long)aligned, n ? 100.0 * aligned / n : 0.0,
            (unsigned long long)(n - aligned), n ? 100.0 * (n - aligned) / n : 0.0, (unsigned long long)aligned);
    mirge_close(c);
    return 0;
}
