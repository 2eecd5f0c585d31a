"""Trimming step (mirge_amd/csrc/trim.h, HOST build) against the CPU restatement of trim_file.py + cutadapt
(oracle/trim_oracle.c).  PARITY UNPINNED with respect to cutadapt itself (not installed, no vector in the reference):
these tests pin the device code to the restatement.  GPU run of the same cases: tests/test_gpu_trim.py."""
import numpy as np
import pytest

from mirge_amd import synth
from oracle import trim as otrim
from tests.test_fastq_cpu import decode

A = synth.ILLUMINA_ADAPTER


def sim_trim(hostsim, data: bytes, adapter: str, trim5=0, trim3=0):
    size = len(data)
    pad = np.full((size + 15) // 16 * 16 + 16, ord("\n"), dtype=np.uint8)
    pad[:size] = np.frombuffer(data, dtype=np.uint8)
    cap = data.count(b"\n") // 4 + 2
    keys = np.zeros((cap, 6), dtype=np.uint64); kw = np.zeros(cap, dtype=np.int32); out = np.zeros(4, dtype=np.int64)
    n = hostsim.sim_trim_fastq(pad.ctypes.data, size, adapter.encode(), trim5, trim3, keys.ctypes.data, kw.ctypes.data, cap, out.ctypes.data)
    if n < 0:
        return n
    rd = [decode(keys[i], int(kw[i])) if kw[i] else data[int(keys[i][0]):int(keys[i][0]) + int(keys[i][1])] for i in range(n)]
    return rd, int(out[0]), int(out[2]), int(out[3])


def raw_case(seed, n=1500, **kw):
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 900 + seed)
    seqs, lens = synth.make_reads(libs, n, 910 + seed)
    return synth.make_raw_fastq(seqs, lens, 920 + seed, **kw)


def test_known_answers():
    ins = b"TGAGGTAGTAGGTTGTATAGTT"
    ab = A.encode()
    assert otrim.adapter_back(ab, ins + ab[:28])[2] == 22                     # the read ends inside the adapter
    assert otrim.adapter_back(ab, ins + ab + b"ACGT")[2:4] == (22, 51)         # whole adapter, leftmost exact occurrence
    assert otrim.adapter_back(ab, ins + b"TGG")[2] == 22                       # 3-nt overlap is the minimum ...
    assert otrim.adapter_back(ab, ins + b"TG") is None                         # ... 2 nt are not removed
    r = bytearray(ins + ab[:28]); r[30] = ord("C") if r[30] != ord("C") else ord("A")
    assert otrim.adapter_back(ab, bytes(r)) == (0, 28, 22, 50, 27, 1)          # one substitution: 1 <= int(28 * 0.12) = 3
    r = bytearray(ins + ab[:8]); r[25] = ord("C") if r[25] != ord("C") else ord("A")
    m = otrim.adapter_back(ab, bytes(r))
    assert m is None or m[5] == 0                                               # 8 nt allow no error: int(8 * 0.12) = 0
    assert otrim.quality_trim(b"IIIIIIIIII####") == (0, 10)
    assert otrim.quality_trim(b"IIII#III##") == (0, 8)                          # BWA: the best cut, not the first low base
    assert otrim.quality_trim(b"####") == (0, 0)


@pytest.mark.parametrize("seed,adapter,t5,t3,kw", [
    (1, A, 0, 0, {}), (2, A, 1, 2, {}), (3, "illumina", 0, 0, {"read_len": 75}), (4, "none", 0, 0, {}), (5, "ion", 0, 0, {}),
    (6, "+11", 2, 0, {}), (7, "-5", 0, 0, {}), (8, A, 0, 0, {"phred": 64}), (9, A, 0, 0, {"crlf": True}),
    (10, "TGGAATTCTCGG," + A, 0, 0, {}), (11, A, 0, 0, {"adapter_error": 0.9, "low_tail": 0.6}), (12, A[:12], 3, 3, {"read_len": 36}),
    (13, A, 0, 0, {"read_len": 124}),
])
def test_host_build_equals_restatement(hostsim, seed, adapter, t5, t3, kw):
    data = raw_case(seed, **kw)
    want, total, base, reported = otrim.trim_fastq(data, {"illumina": A, "ion": "+11"}.get(adapter, adapter), t5, t3)
    got = sim_trim(hostsim, data, adapter, t5, t3)
    assert not isinstance(got, int), got
    assert got[1] == total and got[2] == base and got[3] == reported
    assert got[0] == want
    assert 0 < len(want) <= total
    if adapter not in ("none",) and not adapter.startswith(("+", "-")) and adapter != "ion":
        assert len(set(len(x) for x in want)) > 5          # the adapters were found: lengths spread like the inserts


def test_phred_is_decided_on_the_first_record(hostsim):
    # record 0 all below 'K', a later record above: the workers keep base 33, miRge.pl is told 64 (trim_file.py:123-131)
    rec = lambda q: b"@r\n" + b"ACGTACGTACGTACGTACGT" + b"\n+\n" + q * 20 + b"\n"
    data = rec(b"I") + rec(b"h") + rec(b"I")
    got = sim_trim(hostsim, data, "none")
    assert got[2] == 33 and got[3] == 64
    assert (got[2], got[3]) == otrim.trim_fastq(data, "none")[2:]
    data = rec(b"h") + rec(b"I")
    got = sim_trim(hostsim, data, "none")
    assert got[2] == 64 and got[3] == 64 and got[0] == otrim.trim_fastq(data, "none")[0]


def test_errors(hostsim):
    assert sim_trim(hostsim, b"@r\nACGTACGTACGTACGTACGT\n+\nIIII\n", "none") == -6        # quality line shorter than the sequence
    assert sim_trim(hostsim, b"@r\n" + b"A" * 1001 + b"\n+\n" + b"I" * 1001 + b"\n", "none") == -5
    assert sim_trim(hostsim, b"@r\nACGTACGTACGTACGTA GT\n+\n" + b"I" * 20 + b"\n", "none") == -4
    # lines no key holds are trimmed on their bytes; what is left is a verbatim read unless it fits a key by then
    assert sim_trim(hostsim, b"@r\n" + b"A" * 125 + b"\n+\n" + b"I" * 125 + b"\n", "none")[0] == [b"A" * 125]
    assert sim_trim(hostsim, b"@r\nACGTACGTACGTACGTAXGT\n+\n" + b"I" * 20 + b"\n", "none")[0] == [b"ACGTACGTACGTACGTAXGT"]
    assert sim_trim(hostsim, b"@r\n" + b"ACGT" * 40 + b"\n+\n" + b"I" * 30 + b"#" * 130 + b"\n", "none")[0] == [b"ACGT" * 7 + b"AC"]
    assert sim_trim(hostsim, b"", "none")[0] == []


@pytest.mark.parametrize("seed,kw", [(31, {"read_len": 150}), (32, {"read_len": 150, "odd": 0.3}), (33, {"read_len": 75, "odd": 0.5}),
                                     (34, {"read_len": 250, "odd": 0.1, "adapter_error": 0.5}), (35, {"read_len": 300, "adapter": "A" * 4})])
def test_lines_no_key_holds(hostsim, seed, kw):
    """Raw lines beyond 124 nt (150-nt runs with the adapter still on) and reads with lower case / IUPAC codes / '.':
    trimmed on their bytes, equal to the restatement; most inserts fit a key once the adapter is off, the rest come out
    as verbatim reads."""
    data = raw_case(seed, n=1200, **kw)
    for adapter, t5, t3 in ((A, 0, 0), (A, 1, 2), ("none", 0, 0), ("-7", 0, 1)):
        want, total, base, reported = otrim.trim_fastq(data, adapter, t5, t3)
        got = sim_trim(hostsim, data, adapter, t5, t3)
        assert not isinstance(got, int), got
        assert (got[0], got[1], got[2], got[3]) == (want, total, base, reported), (adapter, t5, t3)
        assert len(want) > 100


def _fq(reads):
    return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))


def test_adapter_syntax_known_answers(hostsim):
    """What cutadapt's -a parser takes beyond a plain short sequence (trim_file.py:14-23, 41): adapters over 32 nt, more
    than four of them, name=SEQUENCE, lower case / U, and 'ADAPTER$' anchored at the read's 3' end.  Expected values by
    construction; the restatement (oracle/trim_oracle.c) must give the same."""
    rng = np.random.default_rng(77)
    rs = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tolist())
    ins = [rs(22), rs(30), rs(18), rs(41)]
    long_ad, ad64 = rs(40), rs(64)
    # (1) 40-nt and 64-nt adapters: whole, cut by the read's end, with one error
    one_err = bytearray(long_ad); one_err[17] = ord("ACGT"["ACGT".index(chr(one_err[17])) ^ 1])
    reads = [ins[0] + long_ad, ins[1] + long_ad[:25], ins[2] + bytes(one_err) + b"ACGT", ins[3] + ad64, ins[0] + ad64[:50], ins[1]]
    for spec in (long_ad.decode() + "," + ad64.decode(), "x=" + long_ad.decode().lower().replace("t", "u") + ",y=" + ad64.decode()):
        got = sim_trim(hostsim, _fq(reads), spec)
        assert got[0] == [ins[0], ins[1], ins[2], ins[3], ins[0], ins[1]], spec
        assert got[0] == otrim.trim_fastq(_fq(reads), spec)[0]
    # (2) six adapters: the read carrying the fifth is cut where it starts
    ads = [rs(20) for _ in range(6)]
    reads = [ins[0] + ads[4] + b"AAAA", ins[1] + ads[5][:12], ins[2] + ads[0], ins[3]]
    spec = ",".join(a.decode() for a in ads)
    got = sim_trim(hostsim, _fq(reads), spec)
    assert got[0][:3] == [ins[0], ins[1], ins[2]] and got[0] == otrim.trim_fastq(_fq(reads), spec)[0]      # (the adapter-free read may lose a chance 3-nt overlap)
    assert sim_trim(hostsim, _fq(reads), ",".join("ACGTACGTAC" for _ in range(17))) == -1          # more than 16: refused
    assert sim_trim(hostsim, _fq(reads), "A" * 65) == -1 and sim_trim(hostsim, _fq(reads), "ACGTJACGT") == -1
    assert sim_trim(hostsim, _fq(reads), "ACGT{3}ACGT") == -1 and sim_trim(hostsim, _fq(reads), "ACGTACGT...TTGGCCAA") == -1     # braces, linked adapters: refused
    # (3) anchored: removed only when it ends where the read ends
    ad = rs(24)
    reads = [ins[0] + ad, ins[1] + ad + b"ACGTAC", ins[2] + ad[:12], ins[3] + ad[:-1], ins[0] + ad[:10] + b"G" + ad[10:]]
    spec = ad.decode() + "$"
    got = sim_trim(hostsim, _fq(reads), spec)
    assert got[0][0] == ins[0]                      # the whole adapter at the end: removed
    assert got[0][1] == reads[1]                    # followed by more read: an anchored adapter does not match
    assert got[0][2] == reads[2]                    # only a prefix of it at the end: not the whole adapter
    assert got[0][3] == ins[3]                      # whole but for its last base (one deletion <= int(0.12 * 24) = 2 errors): removed
    assert got[0][4] == ins[0]                      # one inserted base inside: removed
    assert got[0] == otrim.trim_fastq(_fq(reads), spec)[0]
    plain = sim_trim(hostsim, _fq(reads), ad.decode())
    assert plain[0][1] == ins[1] and plain[0][2] == ins[2]                 # the same adapter without the anchor takes both
    assert otrim.adapter_back(ad, ins[0] + ad, anchored=True)[2:4] == (len(ins[0]), len(ins[0]) + 24)
    assert otrim.adapter_back(ad, ins[1] + ad + b"ACGTAC", anchored=True) is None


IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
         "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT"}


def test_adapter_wildcards_known_answers(hostsim):
    """Adapters with IUPAC wildcard characters (cutadapt's adapter_wildcards, on by default: trim_file.py:14-23 passes the
    string through unchanged): an adapter N matches any read base, a code matches the bases it stands for, a read N
    matches nothing; no exact-occurrence shortcut.  Expected values by construction, and equal to the restatement."""
    rng = np.random.default_rng(78)
    rs = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tolist())
    ins = [rs(22), rs(30), rs(19), rs(25), rs(33)]
    F = rs(20)
    spec = "NNNN" + F.decode()                                   # a random 4-mer in front of the fixed part
    reads = [ins[0] + b"ACGT" + F + b"TTTT", ins[1] + b"GGCA" + F[:11], ins[2] + b"TNGA" + F, ins[3] + b"CC", ins[4]]
    got = sim_trim(hostsim, _fq(reads), spec)
    assert got[0][0] == ins[0]                      # any four bases + the fixed part: cut where the four start
    assert got[0][1] == ins[1]                      # the read ends inside the fixed part
    assert got[0][2] == ins[2]                      # a read N under an adapter N is a mismatch (1 <= int(24 * 0.12) = 2): still found
    assert got[0][3] == (ins[3] + b"CC")[:-4] and got[0][4] == ins[4][:-4]      # ANY read ends in four bases that match NNNN: cutadapt's known effect of an adapter that starts with N
    assert got[0] == otrim.trim_fastq(_fq(reads), spec)[0]
    assert otrim.adapter_back(spec.encode(), reads[0]) == (0, 24, 22, 46, 24, 0)
    assert otrim.adapter_back(spec.encode(), reads[2])[2:6] == (19, 43, 23, 1)
    # a two-base code: R = A or G costs nothing for either, one error for the others
    ad = F[:8] + b"R" + F[9:]
    ra, rg, rc = (ins[0] + F[:8] + x + F[9:] + b"AC" for x in (b"A", b"G", b"C"))
    assert otrim.adapter_back(ad, ra) == (0, 20, 22, 42, 20, 0) and otrim.adapter_back(ad, rg) == (0, 20, 22, 42, 20, 0)
    assert otrim.adapter_back(ad, rc) == (0, 20, 22, 42, 19, 1)
    got = sim_trim(hostsim, _fq([ra, rg, rc]), ad.decode().lower())
    assert got[0] == [ins[0]] * 3 and got[0] == otrim.trim_fastq(_fq([ra, rg, rc]), ad.decode())[0]
    # X matches nothing; anchored + wildcards
    assert otrim.adapter_back(F[:5] + b"X" + F[6:], ins[0] + F)[5] == 1
    got = sim_trim(hostsim, _fq([ins[0] + b"AC" + F, ins[1] + b"AC" + F + b"G"]), "NN" + F.decode() + "$")
    assert got[0][0] == ins[0]
    assert got[0] == otrim.trim_fastq(_fq([ins[0] + b"AC" + F, ins[1] + b"AC" + F + b"G"]), "NN" + F.decode() + "$")[0]


@pytest.mark.parametrize("seed", [7, 8])
def test_random_wildcard_adapters(hostsim, seed):
    """Randomised: 1-4 adapters of 6-48 characters, a share of them IUPAC codes (N runs among them), some anchored; the
    reads carry one concrete spelling of an adapter, with errors.  Host build of the device code against the restatement."""
    rng = np.random.default_rng(4200 + seed)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 77)
    codes = "ACGT" * 6 + "RYSWKMBDHVN" + "NNNN"
    for _ in range(8):
        nad = int(rng.integers(1, 5))
        ads = ["".join(codes[x] for x in rng.integers(0, len(codes), size=int(rng.integers(6, 49)))) for _ in range(nad)]
        spec = ",".join(a + ("$" if rng.random() < 0.25 else "") for a in ads)
        t5, t3 = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        seqs, lens = synth.make_reads(libs, 300, int(rng.integers(0, 1 << 30)))
        lens = np.minimum(lens, rng.integers(1, 48, size=len(lens))).astype(lens.dtype)
        use = "".join(IUPAC[c][int(rng.integers(0, len(IUPAC[c])))] for c in ads[int(rng.integers(0, nad))])
        data = synth.make_raw_fastq(seqs, lens, int(rng.integers(0, 1 << 30)), adapter=use,
                                    read_len=int(rng.choice([36, 50, 75, 100, 150])) if rng.random() < 0.5 else int(rng.integers(20, 60)) + len(use),
                                    adapter_error=float(rng.random()) * 0.7, low_tail=float(rng.random()) * 0.3, odd=0.1)
        want, total, base, reported = otrim.trim_fastq(data, spec, t5, t3)
        got = sim_trim(hostsim, data, spec, t5, t3)
        assert not isinstance(got, int), (got, spec)
        assert (got[0], got[1], got[2], got[3]) == (want, total, base, reported), (spec, t5, t3)
        assert len(set(len(x) for x in want)) > 5, spec          # the adapters were found: lengths spread like the inserts


@pytest.mark.parametrize("seed", [5, 6])
def test_random_long_anchored_and_many_adapters(hostsim, seed):
    """Randomised like the next test: 1-8 adapters of 3-64 nt, every third anchored, lower-case reads now and then."""
    rng = np.random.default_rng(4100 + seed)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 77)
    for _ in range(8):
        nad = int(rng.integers(1, 9))
        ads = ["".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(3, 65)))) for _ in range(nad)]
        spec = ",".join(a + ("$" if rng.random() < 0.33 else "") for a in ads)
        t5, t3 = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        seqs, lens = synth.make_reads(libs, 300, int(rng.integers(0, 1 << 30)))
        lens = np.minimum(lens, rng.integers(1, 48, size=len(lens))).astype(lens.dtype)
        use = ads[int(rng.integers(0, nad))]
        data = synth.make_raw_fastq(seqs, lens, int(rng.integers(0, 1 << 30)), adapter=use,
                                    read_len=int(rng.choice([36, 50, 75, 100, 150])) if rng.random() < 0.5 else int(rng.integers(20, 60)) + len(use),
                                    adapter_error=float(rng.random()), low_tail=float(rng.random()) * 0.3, odd=0.1)
        want, total, base, reported = otrim.trim_fastq(data, spec, t5, t3)
        got = sim_trim(hostsim, data, spec, t5, t3)
        assert not isinstance(got, int), (got, spec)
        assert (got[0], got[1], got[2], got[3]) == (want, total, base, reported), (spec, t5, t3)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_adapters_lengths_and_cuts(hostsim, seed):
    """Randomised: 1-3 adapters of 3-32 nt (homopolymers now and then), raw read lengths 20-124, inserts cut short so the
    adapter starts anywhere (or the read ends inside it), both quality encodings, error and low-quality-tail rates from 0
    to 1, trim5 / trim3 up to 5.  300 such configurations were run once by hand; ten per seed stay here."""
    rng = np.random.default_rng(4000 + seed)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 77)
    for _ in range(10):
        nad = int(rng.integers(1, 4))
        ads = ["".join("ACGT"[x] for x in rng.integers(0, 4, size=int(rng.integers(3, 33)))) for _ in range(nad)]
        ads = [a[0] * len(a) if rng.random() < 0.1 else a for a in ads]
        adapter = ",".join(ads)
        t5, t3 = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        seqs, lens = synth.make_reads(libs, 300, int(rng.integers(0, 1 << 30)))
        lens = np.minimum(lens, rng.integers(1, 48, size=len(lens))).astype(lens.dtype)
        data = synth.make_raw_fastq(seqs, lens, int(rng.integers(0, 1 << 30)), adapter=ads[int(rng.integers(0, nad))],
                                    read_len=int(rng.choice([20, 36, 50, 75, 100, 124])), phred=int(rng.choice([33, 64])),
                                    adapter_error=float(rng.random()), low_tail=float(rng.random()))
        want, total, base, reported = otrim.trim_fastq(data, adapter, t5, t3)
        got = sim_trim(hostsim, data, adapter, t5, t3)
        assert not isinstance(got, int), (got, adapter)
        assert (got[0], got[1], got[2], got[3]) == (want, total, base, reported), (adapter, t5, t3)
