"""Opportunistic checks against the REAL third-party tools.  Neither bowtie 1.x nor
cutadapt is installed on the build container or the GPU pool, so these skip there; they are what pins rows a4 / f1 the
day the repository meets a machine that has the tools.  Nothing is installed or fetched.

bowtie: the golden species libraries are built with bowtie-build, the reference's five command lines (miRge.pl:327-339)
run over the golden reads round by round exactly as runAnnotationPipeline drives them, and the line miRge would keep
(the LAST line per read, miRge.pl:428-430) is compared with the oracle's choice -- uniques whose best stratum holds one
reference and uniques where it holds several (hypothesis H1 decides) are reported separately.
cutadapt: `cutadapt -q 10 -a ADAPTER -e 0.12 -m 16` (what trim_file.py:32-43, 57 assembles from cutadapt's classes)
against oracle/trim_oracle.c on synthetic raw reads."""
import collections
import os
import re
import shutil
import subprocess

import pytest

from mirge_amd import synth
from tests import util


def _version(exe):
    p = shutil.which(exe)
    if not p:
        return None
    try:
        out = subprocess.run([p, "--version"], capture_output=True, text=True, timeout=30)
        m = re.search(r"(\d+\.\d+(\.\d+)?)", out.stdout + out.stderr)
        return m.group(1) if m else None
    except Exception:
        return None


@pytest.mark.parametrize("name", util.golden_cases())
def test_real_bowtie_agrees_with_the_oracle(tmp_path, oracle, name):
    ver = _version("bowtie")
    if not ver or not ver.startswith("1.") or not shutil.which("bowtie-build"):
        pytest.skip("bowtie 1.x not installed (found: %r)" % ver)
    case = util.golden_case(name)
    libs = {}
    for nm in synth.LIB_NAMES:
        fa = os.path.join(case["species"], nm + ".fa")
        subprocess.run(["bowtie-build", "-q", fa, str(tmp_path / nm)], check=True, capture_output=True, timeout=1800)
        libs[nm] = util.read_fasta(fa)
    reads = [util.read_fastq_seqs(f) for f in case["fastqs"]]
    olibs = [oracle.OracleLib(libs[nm][1]) for nm in synth.LIB_NAMES]
    res = oracle.run_path(olibs, reads, use_index=False, maxlen=1000)
    want = {s: (int(res["ann_round"][u]), int(res["ann_ref"][u]), int(res["ann_amb"][u])) for u, s in enumerate(res["uniques"])}
    todo = [s for s in want if s]                                   # (an empty read cannot be written as FASTA for bowtie)
    rounds = [("mirna", lambda s: len(s) < 26, ["-n", "0"]), ("hairpin", lambda s: len(s) > 25, ["-n", "1"]),
              ("other", lambda s: True, ["-n", "1"]), ("mrna", lambda s: True, ["-n", "0"]),
              ("mirna", lambda s: True, ["-l", "15", "-5", "1", "-3", "2", "-n", "2"])]
    got = {s: (0, -1) for s in todo}
    for r, (lib, ok, opts) in enumerate(rounds):
        fa = tmp_path / "SeqToAnnot.fasta"
        with open(fa, "w") as f:
            for s in todo:
                if ok(s):
                    f.write(">%s\n%s\n" % (s, s))
        out = subprocess.run(["bowtie", "-a", "--best", "--strata", str(tmp_path / lib)] + opts + ["-f", str(fa)],
                             capture_output=True, text=True, timeout=3600, check=True).stdout
        short = [n.split()[0] for n in libs[lib][0]]
        last = {}
        for line in out.split("\n"):
            if line:
                c = line.split("\t")
                last[c[0]] = c[2]                                    # miRge.pl:429-430: later lines overwrite
        for s, ref in last.items():
            got[s] = (r + 1, short.index(ref))
        todo = [s for s in todo if s not in last]
    plain = [s for s in got if not want[s][2]]
    tied = [s for s in got if want[s][2]]
    bad_plain = [(s, got[s], want[s][:2]) for s in plain if got[s] != want[s][:2]]
    bad_tied = [(s, got[s], want[s][:2]) for s in tied if got[s] != want[s][:2]]
    print("bowtie %s, case %s: %d uniques with one best reference (%d differ), %d with several (%d differ: hypothesis H1)" %
          (ver, name, len(plain), len(bad_plain), len(tied), len(bad_tied)))
    assert not bad_plain, bad_plain[:5]          # the matcher's arithmetic: strata, -n / -e / -l / -5 / -3, read N, characters (H2 - H4)
    assert not bad_tied, bad_tied[:5]            # the order bowtie prints ties in (H1)


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"read_len": 75}), (3, {"read_len": 150, "odd": 0.1}), (4, {"phred": 64})])
def test_real_cutadapt_agrees_with_the_restatement(tmp_path, seed, kw):
    ver = _version("cutadapt")
    if not ver:
        pytest.skip("cutadapt not installed")
    from oracle import trim as otrim
    from tests.test_trim_cpu import raw_case
    data = raw_case(seed, n=5000, **kw)
    want, total, base, _ = otrim.trim_fastq(data, synth.ILLUMINA_ADAPTER)
    src, dst = tmp_path / "raw.fastq", tmp_path / "trimmed.fastq"
    src.write_bytes(data)
    subprocess.run(["cutadapt", "-q", "10", "--quality-base", str(base), "-a", synth.ILLUMINA_ADAPTER, "-e", "0.12", "-m", "16",
                    "-o", str(dst), str(src)], check=True, capture_output=True, timeout=1800)
    got = util.read_fastq_seqs(str(dst))
    print("cutadapt %s: %d of %d records kept" % (ver, len(got), total))
    assert collections.Counter(got) == collections.Counter(x.decode() for x in want)
