"""CPU checks of the two drop-in front ends: the bowtie-CLI shim's inspect mode and the Perl XS module."""
import os
import shutil
import subprocess

import pytest

from tests import util

ROOT = util.ROOT


def test_bowtie_inspect_prints_headers():
    exe = os.path.join(ROOT, "tools", "mirge-bowtie-inspect")
    if not os.path.exists(exe):
        pytest.skip("tools/mirge-bowtie not built")
    case = util.golden_case("toy")
    out = subprocess.run([exe, "-n", os.path.join(case["species"], "mirna")], capture_output=True, text=True, check=True).stdout
    names, _ = util.read_fasta(os.path.join(case["species"], "mirna.fa"))
    assert out.split("\n")[:-1] == names                       # whole header lines, as `bowtie-inspect -n` (miRge.pl:457)


def test_perl_module_loads_and_fails_loudly_without_gpu():
    if not shutil.which("perl") or not os.path.exists(os.path.join(ROOT, "perl", "auto", "MirgeHip", "MirgeHip.so")):
        pytest.skip("perl binding not built")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_tools.py")
    r = subprocess.run(["perl", "-I" + os.path.join(ROOT, "perl"), "-e",
                        'use MirgeHip; my $c = eval { MirgeHip::open("/nonexistent", 1, 0) }; print defined($c) ? "opened" : "died: $@";'],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "died: MirgeHip::open" in r.stdout and "no CPU path" in r.stdout, r.stdout + r.stderr


def test_abi_exports_every_declared_symbol():
    """include/mirge_hip.h and the shared library agree (no compute without a GPU)."""
    import re
    from mirge_amd import capi
    hdr = open(os.path.join(ROOT, "include", "mirge_hip.h")).read()
    declared = set(re.findall(r"\b(mirge_[a-z0-9_]+)\s*\(", hdr)) - {"mirge_ctx"}
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.load()
    for s in declared:
        assert hasattr(L, s), s
