"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

from mirge_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def read_fasta(path):
    names, seqs = [], []
    for line in open(path):
        line = line.rstrip("\n").rstrip("\r")
        if line.startswith(">"):
            names.append(line[1:]); seqs.append("")
        elif names:
            seqs[-1] += line
    return names, seqs


def read_fastq_seqs(path):
    """miRge.pl:300-304: line 2 of every 4-line record, chomp'ed."""
    L = open(path).read().split("\n")
    if L and L[-1] == "":
        L.pop()
    return [L[i + 1] if i + 1 < len(L) else "" for i in range(0, len(L), 4)]


class OracleCtx:
    """Test double with capi.Context's surface, served by the CPU oracle.  Lets the host-side
    driver (mirge_amd/driver.py) be checked against the golden fixtures without a GPU.  Lives in
    tests/ on purpose: the product never sees it."""

    def __init__(self, species_dir, n_samples, use_index=False):
        from oracle import oracle as orc
        self.orc = orc
        self.S = n_samples
        self.names, self.short, self.libs = [], [], []
        for nm in synth.LIB_NAMES:
            n, s = read_fasta(os.path.join(species_dir, nm + ".fa"))
            self.names.append(n); self.short.append([x.split()[0] if x.split() else x for x in n])
            self.libs.append(orc.OracleLib(s))
        self.samples = [[] for _ in range(n_samples)]
        self.use_index = use_index

    def lib_size(self, lib): return len(self.names[lib])
    def lib_names(self, lib, short=False): return list(self.short[lib] if short else self.names[lib])

    def collapse_fastq(self, path, sample):
        r = read_fastq_seqs(path); self.samples[sample] += r; return len(r)

    def collapse_seqs(self, reads, sample, lens=None):
        self.samples[sample] += synth.reads_to_list(reads, lens) if lens is not None else list(reads)

    def annotate(self):
        self.res = self.orc.run_path(self.libs, self.samples, use_index=self.use_index, maxlen=capi.MAX_READ_LEN)

    def round_stats(self, r):
        return int(self.res["stats"][r - 1, 0]), int(self.res["stats"][r - 1, 1]), 0.0

    def counts(self):
        return {k: self.res[k] for k in ("quant", "iscan", "cat", "lenhist", "tuniq")}

    def n_unique(self): return len(self.res["uniques"])

    def export_uniques(self):
        U = self.n_unique()
        stride = max([128] + [len(s) for s in self.res["uniques"]])
        seqs = np.zeros((U, stride), dtype=np.uint8); lens = np.zeros(U, dtype=np.uint16)
        for i, s in enumerate(self.res["uniques"]):
            seqs[i, :len(s)] = np.frombuffer(s.encode(), dtype=np.uint8); lens[i] = len(s)
        return seqs, lens, self.res["ann_round"], self.res["ann_ref"], self.res["ann_amb"].astype(np.uint8)

    def export_counts(self):
        u, s = np.nonzero(self.res["counts"])
        return u.astype(np.uint64), s.astype(np.uint32), self.res["counts"][u, s].astype(np.uint32)


def golden_cases():
    return sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d, "expected")))


def golden_case(name):
    d = os.path.join(GOLD, name)
    fq = sorted(f for f in os.listdir(d) if f.endswith(".fastq"))
    exp = os.path.join(d, "expected")
    head = open(os.path.join(exp, "miR.Counts.csv")).readline().rstrip("\n").split(",")[1:]
    return dict(dir=d, species=os.path.join(d, "species"), fastqs=[os.path.join(d, f) for f in fq], names=head,
                expected=exp, stats=json.load(open(os.path.join(exp, "stats.json"))))


def rows(path):
    """(header, sorted data rows) -- mapped/unmapped.csv are compared as row sets (Perl hash order)."""
    L = open(path).read().split("\n")
    return L[0], sorted(x for x in L[1:] if x)


def check_run_against_golden(run, case, outdir):
    """run: a finished mirge_amd.driver.MirgeRun whose tables were written to outdir."""
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):                     # byte-identical (to a reference run whose bowtie was the oracle: parity with bowtie itself unpinned)
        assert open(os.path.join(outdir, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):                        # identical as row sets
        assert rows(os.path.join(outdir, f)) == rows(os.path.join(exp, f)), f
    st = case["stats"]
    assert len(run.uniques()) == st["nUnique"]
    for r in range(5):
        assert run.logHash["annotStats"][r + 1]["readsAligned"] == st["annotStats"][r]["readsAligned"], r
    for i, q in enumerate(st["quantStats"]):
        for k in ("trimmedUniq", "mirnaReads", "mirnaReadsFiltered", "mirnaUniqFiltered", "hairpinReads", "ornaReads",
                  "mrnaReads", "remReads"):
            assert run.logHash["quantStats"][i].get(k) == q[k], (i, k)
        assert run.logHash["quantStats"][i].get("trimmedReads") == q["trimmedReads"]


def load_libs_into(ctx, libs):
    for i, nm in enumerate(synth.LIB_NAMES):
        ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)


def oracle_libs(libs, orc):
    return [orc.OracleLib((libs[nm].text, libs[nm].offs)) for nm in synth.LIB_NAMES]


def gpu_result_dict(ctx):
    """{sequence: (round, ref, tuple(counts))} from a finished capi.Context."""
    seqs, lens, rnd, ref, amb = ctx.export_uniques()
    uid, smp, cnt = ctx.export_counts()
    counts = np.zeros((len(lens), ctx.S), dtype=np.int64)
    np.add.at(counts, (uid.astype(np.int64), smp.astype(np.int64)), cnt.astype(np.int64))
    return {seqs[u, :lens[u]].tobytes().decode(): (int(rnd[u]), int(ref[u]), int(amb[u]), tuple(int(x) for x in counts[u]))
            for u in range(len(lens))}


def oracle_result_dict(res):
    return {s: (int(res["ann_round"][u]), int(res["ann_ref"][u]), int(res["ann_amb"][u]), tuple(int(x) for x in res["counts"][u]))
            for u, s in enumerate(res["uniques"])}


def fits_key(r):
    """does a device-resident key hold the read (A/C/G/T/N, <= 124 nt)?  Everything else is a verbatim read."""
    return len(r) <= 124 and all(c in "ACGTN" for c in r)


def with_verbatim(libs, seqs, lens, seed=5, n=60):
    """The sample as list[str] with reads no key holds mixed in (miRge.pl:306 keys them by their exact string, bowtie's
    parser reads a/c/g/t as the base and anything else as N -- hypothesis H4): lower-case and mixed-case copies of reads
    of the sample, IUPAC codes and '.', fragments of 125 .. 400 nt of the long libraries with 0-2 substitutions on either
    strand, and random long reads; several of them more than once and at both ends of the list, so that every
    contiguous shard of the sample holds some."""
    from mirge_amd import synth
    rng = np.random.default_rng(seed)
    reads = synth.reads_to_list(seqs, lens)
    odd = []
    for i in rng.integers(0, len(reads), size=n):
        r = reads[int(i)]
        if not r:
            continue
        u = rng.random()
        if u < 0.35:
            odd.append(r.lower())
        elif u < 0.5:
            odd.append("".join(c.lower() if rng.random() < 0.5 else c for c in r))
        elif u < 0.75:
            j = int(rng.integers(0, len(r)))
            odd.append(r[:j] + str(rng.choice(list("RYKMSWBDHV.nx-*"))) + r[j + 1:])
        else:
            odd.append(r)
    comp = str.maketrans("ACGT", "TGCA")
    for _ in range(n):
        lib = libs[str(rng.choice(["other", "mrna", "mrna"]))]
        i = int(rng.integers(0, lib.n))
        ref = lib.text[int(lib.offs[i]):int(lib.offs[i + 1])].tobytes().decode()
        if len(ref) < 126 or rng.random() < 0.15:
            odd.append("".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(125, 300)))))
            continue
        L = int(rng.integers(125, min(len(ref), 400) + 1))
        a = int(rng.integers(0, len(ref) - L + 1))
        r = ref[a:a + L]
        for _ in range(int(rng.integers(0, 3))):
            j = int(rng.integers(0, L))
            r = r[:j] + "ACGT"[("ACGT".index(r[j]) + 1 + int(rng.integers(0, 3))) % 4] + r[j + 1:] if r[j] in "ACGT" else r
        if rng.random() < 0.4:
            r = r[::-1].translate(comp)
        if rng.random() < 0.2:
            r = r.lower()
        odd.append(r)
    odd = [r for r in odd if not fits_key(r)]
    k = len(odd) // 2
    return odd[:k] + odd[:k // 2] + reads + odd[k:] + odd[:k // 3]
