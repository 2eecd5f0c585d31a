"""GPU checks of the drop-in front ends: tools/mirge-bowtie (the reference's --bowtie plug-in point) and the
Perl XS binding (perl/MirgeHip.xs), both on the golden cases."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
ROOT = util.ROOT
ROUND_ARGS = [("mirna", ["-n", "0"]), ("hairpin", ["-n", "1"]), ("other", ["-n", "1"]), ("mrna", ["-n", "0"]),
              ("mirna", ["-l", "15", "-5", "1", "-3", "2", "-n", "2"])]          # miRge.pl:333-339


def _run(exe, base, args, fasta):
    cmd = [exe, "-a", "--best", "--strata", "--threads", "4", base] + args + ["-f", fasta]        # miRge.pl:327
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    kept = {}
    for line in r.stdout.split("\n"):
        if line:
            c = line.split("\t")
            kept[c[0]] = c[2]                                   # miRge.pl:429-430: last line wins
    import re
    proc = int(re.search(r"reads processed: (\d+)", r.stderr).group(1))                             # miRge.pl:409
    alg = int(re.search(r"reads with at least one reported alignment: (\d+)", r.stderr).group(1))  # miRge.pl:412
    return kept, proc, alg


@pytest.mark.parametrize("name", ["toy", "synth_tiny"])
def test_bowtie_shim_equals_oracle_cli(tmp_path, name, oracle):
    gpu_exe = os.path.join(ROOT, "tools", "mirge-bowtie")
    orc_exe = os.path.join(ROOT, "oracle", "_build", "orc-bowtie")
    case = util.golden_case(name)
    reads = sorted({s for f in case["fastqs"] for s in util.read_fastq_seqs(f)})
    fa = str(tmp_path / "SeqToAnnot.fasta")
    with open(fa, "w") as f:
        for s in reads:
            f.write(">%s\n%s\n" % (s, s))                      # miRge.pl:385
    for lib, args in ROUND_ARGS:
        base = os.path.join(case["species"], lib)
        got = _run(gpu_exe, base, args, fa)
        exp = _run(orc_exe, base, args, fa)
        assert got == exp, (lib, args)


@pytest.mark.parametrize("name", util.golden_cases())
def test_perl_binding_on_golden(name):
    if not shutil.which("perl") or not os.path.exists(os.path.join(ROOT, "perl", "auto", "MirgeHip", "MirgeHip.so")):
        pytest.skip("perl binding not built")
    case = util.golden_case(name)
    r = subprocess.run(["perl", "-I" + os.path.join(ROOT, "perl"), os.path.join(ROOT, "perl", "t", "hotpath.pl"), case["species"]] +
                       case["fastqs"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.split("\n") if l]
    st = case["stats"]
    assert int([x for x in rows if x[0] == "nUnique"][0][1]) == st["nUnique"]
    for x in rows:
        if x[0] == "round":
            assert int(x[3]) == st["annotStats"][int(x[1]) - 1]["readsAligned"]
        if x[0] == "trimmedReads":
            assert int(x[2]) == st["quantStats"][int(x[1])]["trimmedReads"]
    S = len(case["fastqs"])
    cat = np.array([int(v) for v in [x for x in rows if x[0] == "cat"][0][1:]]).reshape(5, S)
    for i, q in enumerate(st["quantStats"]):
        for k, nm in enumerate(["mirnaReads", "hairpinReads", "ornaReads", "mrnaReads", "remReads"]):
            assert cat[k, i] == (q[nm] or 0)
    # per-unique rows against mapped.csv / unmapped.csv of the reference run
    exp = {}
    for f in ("mapped.csv", "unmapped.csv"):
        for line in open(os.path.join(case["expected"], f)).read().split("\n")[1:]:
            if line:
                c = line.split(",")
                exp[c[0]] = (c[2:7], [int(v) for v in c[7:7 + S]])
    lib_of_round = ["mirna", "hairpin", "other", "mrna", "mirna"]
    names = {l: [n.split()[0] for n in util.read_fasta(os.path.join(case["species"], l + ".fa"))[0]] for l in set(lib_of_round)}
    n_uniq = 0
    for x in rows:
        if x[0] != "uniq":
            continue
        n_uniq += 1
        seq, refs, counts = x[1], [int(v) for v in x[2:7]], [int(v) for v in x[8:8 + S]]
        annot = [names[lib_of_round[r]][refs[r]] if refs[r] >= 0 else "" for r in range(5)]
        assert exp[seq] == (annot, counts), seq
    assert n_uniq == len(exp)


@pytest.mark.parametrize("name", util.golden_cases())
def test_batch_cli_writes_the_reference_tables(tmp_path, name):
    """tools/mirge-annotate (batch mode without Perl): tables equal to the unmodified reference's, stats lines equal to
    its logHash numbers."""
    import re
    case = util.golden_case(name)
    out = str(tmp_path / "out")
    r = subprocess.run([os.path.join(ROOT, "tools", "mirge-annotate"), "--species", case["species"], "--SampleFiles", ",".join(case["fastqs"]),
                        "--SampleNames=" + ",".join(case["names"]), "--outputDir", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(out, f)) == util.rows(os.path.join(exp, f)), f
    aligned = [int(x) for x in re.findall(r"readsAligned (\d+)", r.stdout)]
    assert aligned == [case["stats"]["annotStats"][k]["readsAligned"] for k in range(5)]
    bad = subprocess.run([os.path.join(ROOT, "tools", "mirge-annotate"), "--species", str(tmp_path / "nowhere"), "--SampleFiles", case["fastqs"][0]],
                         capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot find" in bad.stderr


@pytest.mark.parametrize("args", [["--gpus", "1"], ["--gpus", "3", "--host-staged"]])
def test_batch_cli_over_a_group_of_gpus(tmp_path, args):
    """tools/mirge-annotate --gpus N: the same tables through mirge_group_* -- one rank over RCCL, and three logical ranks
    on one device with the exchanges staged through host memory (whole sample files per rank)."""
    import re
    case = util.golden_case("synth_tiny")
    out = str(tmp_path / "out")
    r = subprocess.run([os.path.join(ROOT, "tools", "mirge-annotate"), "--species", case["species"], "--SampleFiles", ",".join(case["fastqs"]),
                        "--SampleNames=" + ",".join(case["names"]), "--outputDir", out] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(out, f)) == util.rows(os.path.join(exp, f)), f
    aligned = [int(x) for x in re.findall(r"readsAligned (\d+)", r.stdout)]
    assert aligned == [case["stats"]["annotStats"][k]["readsAligned"] for k in range(5)]
    assert "merge of the unique tables" in r.stdout


@pytest.mark.parametrize("name", util.golden_cases())
def test_perl_full_run_writes_the_reference_tables(tmp_path, name):
    """perl/t/hotpath.pl --out: the XS call sequence of INTEGRATION.md with ONE bulk export (MirgeHip::export_bulk) instead of
    an XSUB call per unique, then miRNAmerge / filter / the four tables in Perl: equal to the unmodified reference's tables
    (the reference ran with orc-bowtie as its bowtie: parity of the matcher's arithmetic itself is unpinned)."""
    if not shutil.which("perl") or not os.path.exists(os.path.join(ROOT, "perl", "auto", "MirgeHip", "MirgeHip.so")):
        pytest.skip("perl binding not built")
    case = util.golden_case(name)
    out = str(tmp_path / "out")
    r = subprocess.run(["perl", "-I" + os.path.join(ROOT, "perl"), os.path.join(ROOT, "perl", "t", "hotpath.pl"), "--species", case["species"],
                        "--out", out, "--names", ",".join(case["names"])] + case["fastqs"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(out, f)) == util.rows(os.path.join(exp, f)), f
    assert "[hotpath] export_bulk" in r.stderr


def test_perl_full_run_over_a_group_of_gpus(tmp_path):
    """The same Perl run with MirgeHip::group_* (--gpus 2 --host-staged: two logical ranks on the box's one GPU, whole
    sample files per rank, merge + all-reduce inside the library): the reference's tables."""
    if not shutil.which("perl") or not os.path.exists(os.path.join(ROOT, "perl", "auto", "MirgeHip", "MirgeHip.so")):
        pytest.skip("perl binding not built")
    case = util.golden_case("synth_isomir")
    out = str(tmp_path / "out")
    r = subprocess.run(["perl", "-I" + os.path.join(ROOT, "perl"), os.path.join(ROOT, "perl", "t", "hotpath.pl"), "--species", case["species"],
                        "--out", out, "--names", ",".join(case["names"]), "--gpus", "2", "--host-staged"] + case["fastqs"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(out, f)) == util.rows(os.path.join(exp, f)), f
