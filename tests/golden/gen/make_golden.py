#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/<case>/ by running the UNMODIFIED reference
driver (/root/reference/miRge.pl) in the build container.

What this pins: the Perl half of the hot path -- collapse (miRge.pl:291-319), round eligibility
(:374-397), last-line-wins resolution (:416-451), count vectors (:453-494), merge (:496-524),
filter (:526-554) and the table formats (:690-907).  What it does NOT pin: bowtie's arithmetic
(bowtie 1.1.1 is absent; the oracle's bowtie-compatible CLI oracle/_build/orc-bowtie stands in
for it through the reference's own --bowtie plug-in point, miRge.pl:59, 91).

How: CPAN modules missing here (HTML::Table, GD::Graph::*, String::Util) are replaced by the
presentation-only stubs in gen/stubs; `python trim_file.py` and `cutadapt -u` by the pass-through
shims in gen/shims (inputs are already adapter-trimmed).  The reference script is copied to a
temp dir for the run only (its seqLibs path is derived from its own location, miRge.pl:39-43,
and /root/reference is read-only) and removed afterwards; nothing of it enters the repo.

Usage:  python tests/golden/gen/make_golden.py      (needs /root/reference; build container only)
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

GEN = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.dirname(GEN)
ROOT = os.path.dirname(os.path.dirname(GOLD))
REF = "/root/reference/miRge.pl"
sys.path.insert(0, ROOT)
from mirge_amd import synth  # noqa: E402


def run_reference(case_dir, species, fastqs, names=None):
    orc = os.path.join(ROOT, "oracle", "_build", "orc-bowtie")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    tmp = tempfile.mkdtemp(prefix="mirge_golden_")
    try:
        shutil.copy(REF, os.path.join(tmp, "miRge.pl"))
        lib = os.path.join(tmp, "miRge.seqLibs", species)
        shutil.copytree(os.path.join(case_dir, "species"), lib)
        for nm in synth.LIB_NAMES:
            open(os.path.join(lib, nm + ".1.ebwt"), "wb").close()
        env = dict(os.environ)
        env["PATH"] = os.path.join(GEN, "shims") + ":" + env["PATH"]
        env["PERL5LIB"] = os.path.join(GEN, "stubs")
        env["PERL_HASH_SEED"] = "0"
        out = os.path.join(tmp, "out")
        cmd = ["perl", os.path.join(tmp, "miRge.pl"), "--species", species, "--CPU", "1", "--bowtie", orc,
               "--outputDir", out, "--SampleFiles", ",".join(os.path.join(case_dir, f) for f in fastqs)]
        if names:
            cmd += ["--SampleNames", ",".join(names)]
        r = subprocess.run(cmd, env=env, cwd=tmp, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference run failed:\n" + r.stdout + r.stderr)
        exp = os.path.join(case_dir, "expected")
        shutil.rmtree(exp, ignore_errors=True)
        os.makedirs(exp)
        for f in ("mapped.csv", "unmapped.csv", "miR.Counts.csv", "miR.RPM.csv"):
            shutil.copy(os.path.join(out, f), os.path.join(exp, f))
        # sort the hash-ordered tables so regenerated fixtures diff cleanly (compared as row sets anyway)
        for f in ("mapped.csv", "unmapped.csv"):
            p = os.path.join(exp, f)
            L = open(p).read().split("\n")
            # mapped.csv rows end ",\n" (miRge.pl:748-749): the "\n" element is joined by ','
            open(p, "w").write("\n".join([L[0]] + sorted(x for x in L[1:] if x)) + "\n")
        rows = [l.split("\t")[1:] for l in open(os.path.join(out, "report.html")).read().split("\n")
                if l.startswith("GOLDENROW")]
        stats = {"quantStats": [], "annotStats": [], "nUnique": None}
        strip = lambda s: re.sub(r"<[^>]*>", "", s).replace("&nbsp;", "")
        for row in rows:
            if row[0] in ("File name&#40;s&#41;", "Annotation-Round", "Files") or row[0].startswith("<a "):
                continue
            if row[0] == "all sequences":
                stats["nUnique"] = int(row[1])
            elif len(row) == 3:
                stats["annotStats"].append({"round": row[0], "readsAligned": int(row[1]) if row[1] != "" else None})
            elif len(row) == 10:
                tr, tu = strip(row[2]).split("/")
                mr, mf = strip(row[3]).split("/")
                num = lambda s: int(s) if s.strip() != "" else None
                stats["quantStats"].append({"name": row[0], "totalReads": num(row[1]), "trimmedReads": num(tr),
                                            "trimmedUniq": num(tu), "mirnaReads": num(mr), "mirnaReadsFiltered": num(mf),
                                            "mirnaUniqFiltered": num(row[4]), "hairpinReads": num(row[5]),
                                            "ornaReads": num(row[6]), "mrnaReads": num(row[7]), "remReads": num(row[8])})
        json.dump(stats, open(os.path.join(exp, "stats.json"), "w"), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def write_fastq(path, reads):
    with open(path, "w") as f:
        for i, s in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def mut(s, pos, rng=None):
    c = "ACGT"["ACGT".index(s[pos]) ^ 1]     # A<->C, G<->T
    return s[:pos] + c + s[pos + 1:]


def case_toy():
    """Hand-built case: every feature SURVEY.md §8c lists."""
    d = os.path.join(GOLD, "toy")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "species"))
    rng = np.random.default_rng(20141)
    rs = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    mir = {}
    mir["tst-miR-1"] = rs(22)
    mir["tst-miR-2"] = rs(23)
    mir["tst-miR-long"] = rs(28)                     # only windows >= 26 nt possible: never round 1
    mir["tst-miR-4"] = rs(22)
    mir["tst-miR-4.SNPC"] = mut(mir["tst-miR-4"], 10)
    mir["tst-miR-7a"] = rs(22)
    mir["tst-miR-7b"] = mut(mir["tst-miR-7a"], 18)
    mir["tst-miR-9"] = rs(24)                        # isomiR-only: iscan stays 0 -> filtered to 0
    mir["tst-miR-10"] = rs(22)                       # a single exact read: iscan 1 < 2 -> filtered to 0
    mir["tst-miR-11"] = rs(23)                       # never seen: stays 0
    mir["tst-miR-12-5p extra header words"] = rs(22)  # header with spaces (bowtie prints first word only)
    hp = {}
    hp["tst-mir-1"] = rs(20) + mir["tst-miR-1"] + rs(30)
    hp["tst-mir-long"] = rs(15) + mir["tst-miR-long"] + rs(25)
    hp["tst-mir-x"] = rs(80)
    hp["tst-mir-y"] = rs(70)
    oth = {"tRNA-Gly-1": rs(75), "snoRNA-U3": rs(140), "rRNA-5S": rs(120), "ncRNA-withN": rs(40) + "NN" + rs(40)}
    mrn = {"NM_0001": rs(600), "NM_0002": rs(900), "NM_0003": rs(400)}
    merges = [["tst-miR-7a/7b", "tst-miR-7a", "tst-miR-7b"], ["tst-miR-4", "tst-miR-4.SNPC", ""]]
    for nm, lib in zip(synth.LIB_NAMES, (mir, hp, oth, mrn)):
        synth.Library.from_seqs(list(lib.keys()), list(lib.values())).write_fasta(os.path.join(d, "species", nm + ".fa"))
    with open(os.path.join(d, "species", "merges.csv"), "w") as f:
        for row in merges:
            f.write(",".join(row) + "\n")
    m = mir
    s1 = []
    s1 += [m["tst-miR-1"]] * 7                       # exact '+'
    s1 += [rc(m["tst-miR-1"])] * 2                   # exact '-' counts the same (no --norc)
    s1 += [m["tst-miR-1"][:20]] * 3                  # exact sub-window, len < 26
    s1 += [m["tst-miR-2"]] * 5
    s1 += [m["tst-miR-2"] + "A"] * 2                 # 3' addition -> isomiR (round 5)
    s1 += [m["tst-miR-long"][:25]] * 4               # len 25: round 1 eligible, exact
    s1 += [m["tst-miR-long"][:26]] * 3               # len 26: NOT round 1; hairpin exact (round 2)
    s1 += [m["tst-miR-4"]] * 2 + [m["tst-miR-4.SNPC"]] * 3
    s1 += [m["tst-miR-7a"]] * 4 + [m["tst-miR-7b"]] * 1
    s1 += [mut(m["tst-miR-9"], 12)] * 6              # 1 mm -> isomiR only
    s1 += [m["tst-miR-10"]] * 1
    s1 += [m["tst-miR-12-5p extra header words"]] * 2
    s1 += [hp["tst-mir-x"][10:40]] * 2               # hairpin exact, len 30
    s1 += [mut(hp["tst-mir-y"][5:37], 9)] * 1        # hairpin 1 mm in seed
    s1 += [oth["tRNA-Gly-1"][3:33]] * 4 + [mut(oth["snoRNA-U3"][50:70], 4)] * 2
    s1 += [oth["ncRNA-withN"][25:55]] * 1            # overlaps reference N: not alignable there
    s1 += [mrn["NM_0001"][100:135]] * 2 + [rc(mrn["NM_0002"][300:320])] * 1
    s1 += [mut(mrn["NM_0003"][10:40], 3)] * 1        # mRNA needs exact seed: unaligned in round 4
    s1 += [mut(mut(mrn["NM_0003"][50:95], 30), 40)] * 1   # 2 mm beyond nt 28 are fine in round 4
    s1 += [rs(30), rs(18), rs(47)]
    s1 += [m["tst-miR-1"][:10] + "N" + m["tst-miR-1"][11:]] * 2   # read N: 1 mm -> isomiR
    s2 = []
    s2 += [m["tst-miR-1"]] * 3 + [m["tst-miR-2"]] * 1 + [m["tst-miR-7b"]] * 5
    s2 += [m["tst-miR-10"]] * 1 + [m["tst-miR-9"][1:]] * 2 + [mut(m["tst-miR-9"], 12)] * 1
    s2 += [oth["rRNA-5S"][20:60]] * 9 + [mrn["NM_0002"][5:50]] * 2
    s2 += [rs(25), rs(26)]
    s3 = [rs(20), rs(33), oth["rRNA-5S"][0:30], mrn["NM_0001"][200:230]]     # no miRNA at all: empty miRNAtotal cell
    rng.shuffle(s1); rng.shuffle(s2)
    write_fastq(os.path.join(d, "s1.fastq"), s1)
    write_fastq(os.path.join(d, "s2.fastq"), s2)
    write_fastq(os.path.join(d, "s3.fastq"), s3)
    run_reference(d, "toy", ["s1.fastq", "s2.fastq", "s3.fastq"], names=["S1", "S2", "S3"])


def case_synth(name, profile, n_per_sample, n_samples, seed, mixture=None):
    d = os.path.join(GOLD, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    libs, merges = synth.make_libraries(profile, seed)
    synth.write_species_dir(os.path.join(d, "species"), libs, merges, with_dummy_ebwt=False)
    fq = []
    for s in range(n_samples):
        seqs, lens = synth.make_reads(libs, n_per_sample, seed + 100 + s, mixture)
        synth.write_fastq(os.path.join(d, "s%d.fastq" % s), seqs, lens)
        fq.append("s%d.fastq" % s)
    run_reference(d, name, fq)


def case_long(name="synth_long", seed=synth.BASE_SEED + 21):
    """Reads the trimmed-read mixtures never hold: 5 .. 100 nt (the 48-byte key class for everything over 56 nt, and
    reads shorter than a seed piece), fragments of every library with 0-2 substitutions, either strand, some N."""
    d = os.path.join(GOLD, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    libs, merges = synth.make_libraries(synth.Profile.tiny(), seed)
    synth.write_species_dir(os.path.join(d, "species"), libs, merges, with_dummy_ebwt=False)
    rng = np.random.default_rng(seed)
    fq = []
    for s in range(2):
        reads = []
        for _ in range(900):
            u = rng.random()
            if u < 0.2:
                r = "".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(5, 101))))
            else:
                lib = libs[str(rng.choice(["mirna", "hairpin", "other", "other", "mrna", "mrna"]))]
                i = int(rng.integers(0, lib.n))
                ref = lib.text[int(lib.offs[i]):int(lib.offs[i + 1])].tobytes().decode()
                L = min(len(ref), int(rng.integers(8, 101)))
                a = int(rng.integers(0, len(ref) - L + 1))
                r = ref[a:a + L]
                for _ in range(int(rng.integers(0, 3))):
                    r = mut(r, int(rng.integers(0, len(r))), rng)
                if rng.random() < 0.3:
                    r = rc(r)
            if rng.random() < 0.03:
                j = int(rng.integers(0, len(r)))
                r = r[:j] + "N" + r[j + 1:]
            reads.append(r)
        reads += reads[:150]                       # duplicates within the sample
        rng.shuffle(reads)
        write_fastq(os.path.join(d, "s%d.fastq" % s), reads)
        fq.append("s%d.fastq" % s)
    run_reference(d, name, fq)


def case_verbatim(name="synth_verbatim", seed=synth.BASE_SEED + 31):
    """Reads no 2-bit key holds (miRge.pl:306 keys the hash by whatever line 2 holds): 125 .. 400 nt, lower and mixed case,
    IUPAC codes, '.', 'x', '-', '*' -- among ordinary reads, some in both samples.  orc-bowtie reads a/c/g/t as the base and
    every other character as N (hypothesis H4)."""
    from tests import util
    d = os.path.join(GOLD, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    libs, merges = synth.make_libraries(synth.Profile.tiny(), seed)
    synth.write_species_dir(os.path.join(d, "species"), libs, merges, with_dummy_ebwt=False)
    fq = []
    for s in range(2):
        seqs, lens = synth.make_reads(libs, 1000, seed + 100 + s)
        reads = util.with_verbatim(libs, seqs, lens, seed=seed + (0 if s == 0 else 0), n=80)   # the same odd reads in both samples
        write_fastq(os.path.join(d, "s%d.fastq" % s), reads)
        fq.append("s%d.fastq" % s)
    run_reference(d, name, fq)


CASES = {"toy": case_toy,
         "synth_tiny": lambda: case_synth("synth_tiny", synth.Profile.tiny(), 1500, 3, synth.BASE_SEED + 7),
         "synth_isomir": lambda: case_synth("synth_isomir", synth.Profile.tiny(), 1200, 2, synth.BASE_SEED + 11, synth.Mixture.isomir_rich()),
         "synth_long": case_long, "synth_verbatim": case_verbatim}

if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    for nm in (sys.argv[1:] or list(CASES)):          # no arguments: every case
        CASES[nm]()
    print("golden fixtures written under", GOLD)
