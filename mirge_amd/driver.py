"""Host side of the hot path, mirroring the reference driver's own subs (same names, same
argument meaning, same outputs) on top of the C ABI.  The reference is a Perl script
(/root/reference/miRge.pl); perl/MirgeHip.xs + INTEGRATION.md show the binding a maintainer adds
there.  This Python mirror exists so the parity tests read like the reference: every method
cites the sub it stands for.

Only the bookkeeping the reference does in Perl after the heavy lifting (merges.csv folding,
the iscan filter, CSV formatting -- SURVEY.md rows a9-a12, "tiny, stays on host") is computed
here; collapse, cascade and count vectors come from the device through `capi.Context`.
"""
from __future__ import annotations

import os
import numpy as np

from . import capi

ANNOT_NAMES = ["exact miRNA", "hairpin miRNA", "non miRNA/mRNA RNA", "mRNA", "isomiR miRNA"]   # miRge.pl:111
ROUND_LIBS = [capi.LIB_MIRNA, capi.LIB_HAIRPIN, capi.LIB_OTHER, capi.LIB_MRNA, capi.LIB_MIRNA]    # miRge.pl:333-339


def perl_num(x) -> str:
    """Perl's default stringification of a number (%.15g)."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    s = "%.15g" % x
    return s


class MirgeRun:
    """State of one miRge run: $seqHash / $mirHash / $logHash of miRge.pl:99-104, device-backed."""

    def __init__(self, species_dir, sample_files, sample_names=None, device=0, ctx=None, adapter=None, trim5=0, trim3=0):
        # adapter (miRge.pl:80-86 `--adapter`): None = the sample files are already trimmed (the hot path proper);
        # "none" / "illumina" / "ion" / "+N" / a sequence = they are raw and go through the device trimming step
        self.adapter, self.trim5, self.trim3 = adapter, int(trim5), int(trim3)
        self.sampleFiles = list(sample_files)
        self.sampleNames = list(sample_names) if sample_names else list(sample_files)   # miRge.pl:73-78
        self.S = len(self.sampleFiles)
        self.species_dir = species_dir
        self.ctx = ctx if ctx is not None else capi.Context(species_dir, self.S, device)
        self.logHash = {"quantStats": [dict() for _ in range(self.S)], "annotStats": [None] + [dict() for _ in range(5)]}
        self.mirHash = {}
        self._uniq = None

    # ---- miRge.pl:291-319
    def quantReads(self, infile, sampleIndex):
        """Collapse one sample's trimmed FASTQ into the shared unique table."""
        n = self.ctx.collapse_fastq(infile, sampleIndex)
        self.logHash["quantStats"][sampleIndex]["trimmedReads"] = n
        return n

    # ---- miRge.pl:259-262: trimRaw (python trim_file.py) + `cutadapt -u` + quantReads in one device pass
    def trimAndQuantReads(self, infile, sampleIndex):
        total, kept, phred = self.ctx.collapse_raw_fastq(infile, sampleIndex, self.adapter, self.trim5, self.trim3)
        q = self.logHash["quantStats"][sampleIndex]
        q["totalReads"], q["trimmedReads"] = total, kept       # miRge.pl:285-288
        self.phred64 = phred == 64                             # miRge.pl:278
        return kept

    def quantReadsFromList(self, reads, sampleIndex, lens=None):
        self.ctx.collapse_seqs(reads, sampleIndex, lens)

    # ---- miRge.pl:321-372
    def runAnnotationPipeline(self):
        """The five rounds (writeSeqToAnnot -> bowtie -> parseBowtieLog/parseAlignment ->
        updateAnnotHash), in one device pass."""
        self.ctx.annotate()
        for i in range(5):
            p, a, sec = self.ctx.round_stats(i + 1)
            self.logHash["annotStats"][i + 1] = {"readsProcessed": p, "readsAligned": a, "cpuTime": sec}
        self._uniq = None

    # ---- miRge.pl:453-494
    def summarize(self):
        ctx = self.ctx
        full = ctx.lib_names(capi.LIB_MIRNA)                   # `bowtie-inspect -n`, :457-463
        short = ctx.lib_names(capi.LIB_MIRNA, short=True)      # what bowtie prints as the reference name
        self.mirHash = {nm: {"quant": [None] * self.S, "iscan": [None] * self.S} for nm in full}
        cv = ctx.counts()
        self._countvec = cv
        q, isc = cv["quant"], cv["iscan"]
        for m in np.nonzero(q.sum(axis=1))[0]:
            e = self.mirHash.setdefault(short[m], {"quant": [None] * self.S, "iscan": [None] * self.S})
            for s in range(self.S):
                if q[m, s]:
                    e["quant"][s] = (e["quant"][s] or 0) + int(q[m, s])        # :473 / :476
                if isc[m, s]:
                    e["iscan"][s] = (e["iscan"][s] or 0) + int(isc[m, s])      # :474
        keys = ["mirnaReads", "hairpinReads", "ornaReads", "mrnaReads", "remReads"]
        for s in range(self.S):
            st = self.logHash["quantStats"][s]
            st["trimmedUniq"] = int(cv["tuniq"][s]) or None                    # :468
            for k, name in enumerate(keys):
                st[name] = int(cv["cat"][k, s]) or None                        # :470-489 (undef unless incremented)

    # ---- miRge.pl:496-524
    def miRNAmerge(self, merge_file=None):
        path = merge_file or os.path.join(self.species_dir, "merges.csv")
        if not os.path.exists(path):
            return False                                                       # :521-523
        with open(path) as fh:
            for line in fh:
                line = line[:-1] if line.endswith("\n") else line              # chomp
                cols = line.split(",")
                while cols and cols[-1] == "":                                 # Perl split drops trailing empty fields
                    cols.pop()
                if not cols:
                    continue
                for name in cols[1:]:
                    src = self.mirHash.get(name)
                    if src is not None:
                        for j in range(self.S):
                            if (src["quant"][j] or 0) > 0:                     # :511
                                tgt = self.mirHash.setdefault(cols[0], {"quant": [None] * self.S, "iscan": [None] * self.S})
                                tgt["quant"][j] = (tgt["quant"][j] or 0) + src["quant"][j]            # :512
                                tgt["iscan"][j] = (tgt["iscan"][j] or 0) + (src["iscan"][j] or 0)     # :513
                    self.mirHash.pop(name, None)                               # :516
        return True

    # ---- miRge.pl:526-554
    def filter(self):
        for e in self.mirHash.values():
            for i in range(self.S):
                if (e["iscan"][i] or 0) < 2:                                   # :534
                    e["quant"][i] = 0
        for e in self.mirHash.values():
            for i in range(self.S):
                if (e["quant"][i] or 0) > 0:                                   # :542-545
                    st = self.logHash["quantStats"][i]
                    st["mirnaReadsFiltered"] = (st.get("mirnaReadsFiltered") or 0) + e["quant"][i]
                    st["mirnaUniqFiltered"] = (st.get("mirnaUniqFiltered") or 0) + 1

    # ---- per-unique view ($seqHash), miRge.pl:99-101
    def uniques(self):
        """list of (sequence, annot[0..5] as the reference stores them, counts[S])."""
        if self._uniq is not None:
            return self._uniq
        seqs, lens, rnd, ref, amb = self.ctx.export_uniques()
        uid, smp, cnt = self.ctx.export_counts()
        counts = np.zeros((len(lens), self.S), dtype=np.int64)
        np.add.at(counts, (uid.astype(np.int64), smp.astype(np.int64)), cnt.astype(np.int64))
        names = [self.ctx.lib_names(l, short=True) for l in range(4)]
        out = []
        for u in range(len(lens)):
            s = seqs[u, :lens[u]].tobytes().decode()
            annot = [0, None, None, None, None, None]
            if rnd[u] > 0:
                annot[0] = 1
                annot[rnd[u]] = names[ROUND_LIBS[rnd[u] - 1]][ref[u]]           # :445-446
            out.append((s, annot, counts[u]))
        self._uniq = out
        self._amb = amb
        return out

    # ---- miRge.pl:690-907 (mapped.csv :702-752, unmapped.csv :845-863, miR.Counts.csv :865-885,
    #      miR.RPM.csv :887-906)
    def writeDataToCSV(self, outputPath):
        os.makedirs(outputPath, exist_ok=True)
        head = "uniqueSequence,annotFlag," + ",".join(ANNOT_NAMES) + "".join("," + n for n in self.sampleNames) + "\n"
        uq = self.uniques()
        with open(os.path.join(outputPath, "mapped.csv"), "w") as fh:
            fh.write(head)
            for s, annot, counts in uq:
                if annot[0] > 0:
                    entry = [s] + ["" if a is None else str(a) for a in annot] + [str(int(c)) for c in counts] + ["\n"]
                    fh.write(",".join(entry))                                  # :748-749 => rows end ",\n"
        with open(os.path.join(outputPath, "unmapped.csv"), "w") as fh:
            fh.write(head)
            for s, annot, counts in uq:
                if annot[0] == 0:
                    fh.write(s + "".join("," + ("" if a is None else str(a)) for a in annot) +
                             "".join("," + str(int(c)) for c in counts) + "\n")
        tot = [self.logHash["quantStats"][i].get("mirnaReadsFiltered") for i in range(self.S)]
        with open(os.path.join(outputPath, "miR.Counts.csv"), "w") as fh:
            fh.write("miRNA" + "".join("," + n for n in self.sampleNames) + "\n")
            fh.write("miRNAtotal" + "".join("," + ("" if t is None else str(t)) for t in tot) + "\n")   # :874 undef => empty
            for k in sorted(self.mirHash, key=lambda x: x.encode()):                                   # :878 bytewise sort
                fh.write(k + "".join("," + str(self.mirHash[k]["quant"][i] or 0) for i in range(self.S)) + "\n")
        with open(os.path.join(outputPath, "miR.RPM.csv"), "w") as fh:
            fh.write("miRNA" + "".join("," + n for n in self.sampleNames) + "\n")
            for k in sorted(self.mirHash, key=lambda x: x.encode()):
                cells = []
                for i in range(self.S):
                    if tot[i]:                                                                          # :897
                        cells.append(perl_num(1000000 * (self.mirHash[k]["quant"][i] or 0) / tot[i]))
                    else:
                        cells.append("0")
                fh.write(k + "".join("," + c for c in cells) + "\n")

    def run(self, outputPath):
        """quantReads per sample -> runAnnotationPipeline -> summarize -> miRNAmerge -> filter ->
        writeDataToCSV: miRge.pl:133-150 without trimming, graphs and HTML."""
        for i, f in enumerate(self.sampleFiles):
            if self.adapter is None:
                self.quantReads(f, i)
            else:
                self.trimAndQuantReads(f, i)
        self.runAnnotationPipeline()
        self.summarize()
        self.miRNAmerge()
        self.filter()
        self.writeDataToCSV(outputPath)
