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
        seqs = np.zeros((U, 128), dtype=np.uint8); lens = np.zeros(U, dtype=np.uint8)
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
    for f in ("miR.Counts.csv", "miR.RPM.csv"):                     # byte-identical
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
