"""Deterministic synthetic seqLibs + adapter-trimmed reads.

The real miRge.seqLibs (681 MB download, reference docs/index.html:36) are not
available offline, so the bench and the parity tests run on synthetic
libraries with a human-like size profile and on reads drawn from them with
the mixture SURVEY.md §8d states.  Everything is a pure function of the seed.

Reads are produced already adapter-trimmed (the hot path of BASELINE.json
starts there; trimming is reference trim_file.py:32-57, out of scope) as a
fixed-width ASCII matrix `seqs[n, W]` (zero padded) plus `lens[n]`.
"""
from __future__ import annotations

import dataclasses
import numpy as np

BASE_SEED = 0x6D69526765  # "miRge"
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
LIB_NAMES = ("mirna", "hairpin", "other", "mrna")


@dataclasses.dataclass
class Library:
    names: list            # full FASTA header lines (no '>')
    text: np.ndarray       # uint8 ASCII, concatenated, no separators
    offs: np.ndarray       # int64[n+1]

    @property
    def n(self) -> int:
        return len(self.names)

    def seq(self, i: int) -> str:
        return self.text[self.offs[i]:self.offs[i + 1]].tobytes().decode()

    def write_fasta(self, path: str) -> None:
        with open(path, "w") as f:
            for i, nm in enumerate(self.names):
                f.write(">%s\n%s\n" % (nm, self.seq(i)))

    @staticmethod
    def from_seqs(names, seqs) -> "Library":
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        text = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
        return Library(list(names), text, offs)


@dataclasses.dataclass
class Profile:
    """Library size profile.  `human()` is SURVEY.md §8d's default."""
    n_mirna: int = 2700
    n_snp: int = 40            # ".SNPx" near-duplicates of existing entries
    n_family: int = 15         # family groups of 2-4 entries differing by 1-2 nt
    n_hairpin: int = 1900
    n_other: int = 20000
    other_median: float = 150.0
    n_mrna: int = 40000
    mrna_median: float = 2500.0
    mrna_cap: int = 20000
    prefix: str = "hsa"
    # repeat-rich libraries: `mrna` as isoform families (each gene 2-6 transcripts sharing >= 70 % of its exons), `other`
    # with tRNA-like families (10-50 members differing by 1-3 nt) and a few low-complexity stretches.  The default
    # profile draws both libraries as i.i.d. uniform bases: no k-mer repeats, the best case for
    # direct-addressed slot tables.
    repeats: bool = False

    @staticmethod
    def human() -> "Profile":
        return Profile()

    @staticmethod
    def human_repeats() -> "Profile":
        return Profile(repeats=True)

    @staticmethod
    def small_repeats() -> "Profile":
        return Profile(n_mirna=300, n_snp=10, n_family=6, n_hairpin=200, n_other=900, other_median=120.0,
                       n_mrna=300, mrna_median=1500.0, mrna_cap=8000, repeats=True)

    @staticmethod
    def mouse() -> "Profile":
        return Profile(n_mirna=2000, n_hairpin=1200, prefix="mmu")

    @staticmethod
    def tiny() -> "Profile":
        return Profile(n_mirna=40, n_snp=4, n_family=3, n_hairpin=24, n_other=30, other_median=120.0,
                       n_mrna=12, mrna_median=900.0, mrna_cap=3000)

    @staticmethod
    def small() -> "Profile":
        return Profile(n_mirna=300, n_snp=10, n_family=6, n_hairpin=200, n_other=600, other_median=150.0,
                       n_mrna=300, mrna_median=1500.0, mrna_cap=8000)

    @staticmethod
    def medium() -> "Profile":
        return Profile(n_mirna=2700, n_hairpin=1900, n_other=5000, n_mrna=4000, mrna_median=2000.0)


def _rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)]


def _mutate(rng, s, k):
    s = s.copy()
    pos = rng.choice(len(s), size=k, replace=False)
    for p in pos:
        cur = int(np.searchsorted(ACGT, s[p]))
        s[p] = ACGT[(cur + int(rng.integers(1, 4))) % 4]
    return s


def make_libraries(profile: Profile, seed: int = BASE_SEED):
    """Returns ({lib name: Library}, merges rows [[target, src1, ...], ...])."""
    rng = np.random.default_rng(seed)
    P = profile
    # ---- mirna: base entries, then family variants and SNP near-duplicates
    n_base = max(1, P.n_mirna - P.n_snp - 3 * P.n_family)
    names, seqs = [], []
    for i in range(n_base):
        names.append("%s-miR-%d" % (P.prefix, i + 1))
        seqs.append(_rand_seq(rng, int(rng.integers(22, 32))))
    merges = []
    fam_src = rng.choice(n_base, size=min(P.n_family, n_base), replace=False)
    for b in fam_src:
        k = int(rng.integers(1, 4))
        members = [names[b]]
        for j in range(k):
            nm = "%s%s" % (names[b], "bcd"[j])
            names.append(nm); seqs.append(_mutate(rng, seqs[b], int(rng.integers(1, 3))))
            members.append(nm)
        merges.append(["/".join(members)] + members)            # brand-new target name
    snp_src = rng.choice(n_base, size=min(P.n_snp, n_base), replace=False)
    done = {}
    for b in snp_src:
        letter = "CAB"[done.get(b, 0) % 3]; done[b] = done.get(b, 0) + 1
        nm = "%s.SNP%s" % (names[b], letter)
        names.append(nm); seqs.append(_mutate(rng, seqs[b], 1))
        merges.append([names[b], nm])                           # fold into the existing name
    order = rng.permutation(len(names))
    names = [names[i] for i in order]; seqs = [seqs[i] for i in order]
    mirna = Library.from_seqs(names, [s.tobytes().decode() for s in seqs])
    # ---- hairpin: each embeds 1-2 mirna entries
    hn, hs = [], []
    for i in range(P.n_hairpin):
        L = int(rng.integers(60, 121))
        s = _rand_seq(rng, L)
        a = mirna.text[mirna.offs[i % mirna.n]:mirna.offs[i % mirna.n + 1]]
        p = int(rng.integers(0, 12)); s[p:p + len(a)] = a[:L - p]
        if rng.random() < 0.5:
            j = int(rng.integers(0, mirna.n))
            b = mirna.text[mirna.offs[j]:mirna.offs[j + 1]]
            q = L - len(b) - int(rng.integers(0, 8))
            if q > p + len(a):
                s[q:q + len(b)] = b
        hn.append("%s-mir-%d" % (P.prefix, i + 1)); hs.append(s.tobytes().decode())
    hairpin = Library.from_seqs(hn, hs)

    def lognormal_lib(n, median, cap, lo, tag):
        lens = np.clip(np.exp(rng.normal(np.log(median), 0.6, size=n)), lo, cap).astype(np.int64)
        offs = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=offs[1:])
        text = ACGT[rng.integers(0, 4, size=int(offs[-1]))]
        return Library(["%s_%06d" % (tag, i) for i in range(n)], text, offs)

    def family_lib(n, median, tag):
        """~60 % of the entries in families of 10-50 members that differ from their founder by 1-3 substitutions (tRNA-like:
        every k-mer of a founder occurs 10-50 times), ~1 % of all entries with a low-complexity stretch (a homopolymer or
        a dinucleotide repeat of 30-60 nt: one k-mer hundreds of times), the rest unrelated."""
        seqs_ = []
        while len(seqs_) < int(0.6 * n):
            founder = _rand_seq(rng, int(np.clip(np.exp(rng.normal(np.log(median * 0.6), 0.2)), 60, 200)))
            for _ in range(int(rng.integers(10, 51))):
                seqs_.append(_mutate(rng, founder, int(rng.integers(1, 4))))
        seqs_ = seqs_[:int(0.6 * n)]
        while len(seqs_) < n:
            seqs_.append(_rand_seq(rng, int(np.clip(np.exp(rng.normal(np.log(median), 0.6)), 50, 5000))))
        for i in rng.choice(n, size=max(1, n // 100), replace=False):
            L = int(rng.integers(30, 61))
            unit = [b"A", b"T", b"CA", b"GT", b"AT"][int(rng.integers(0, 5))]
            rep = np.frombuffer((unit * L)[:L], dtype=np.uint8)
            if len(seqs_[i]) > L + 10:
                p = int(rng.integers(0, len(seqs_[i]) - L))
                seqs_[i] = seqs_[i].copy(); seqs_[i][p:p + L] = rep
        order_ = rng.permutation(n)
        lens = np.array([len(seqs_[i]) for i in order_], dtype=np.int64)
        offs = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=offs[1:])
        return Library(["%s_%06d" % (tag, i) for i in range(n)], np.concatenate([seqs_[i] for i in order_]), offs)

    def isoform_lib(n, median, cap, tag):
        """Genes of 4-16 exons; 2-6 transcripts per gene, each keeping >= 70 % of the gene's exons (the first and last
        always): most k-mers of the library occur once per isoform of their gene."""
        parts, lens = [], []
        while len(lens) < n:
            target = float(np.clip(np.exp(rng.normal(np.log(median), 0.6)), 200, cap))
            nex = int(rng.integers(4, 17))
            cut = np.sort(rng.random(nex - 1)); w = np.diff(np.concatenate([[0.0], cut, [1.0]]))
            exons = [_rand_seq(rng, max(30, int(target * 1.15 * x))) for x in w]
            for _ in range(int(rng.integers(2, 7))):
                if len(lens) >= n:
                    break
                keep = rng.random(nex) < 0.85
                keep[0] = keep[-1] = True
                while keep.sum() < int(np.ceil(0.7 * nex)):
                    keep[int(rng.integers(0, nex))] = True
                t = np.concatenate([e for e, k in zip(exons, keep) if k])[:cap]
                parts.append(t); lens.append(len(t))
        offs = np.zeros(n + 1, dtype=np.int64); np.cumsum(np.array(lens, dtype=np.int64), out=offs[1:])
        return Library(["%s_%06d" % (tag, i) for i in range(n)], np.concatenate(parts), offs)

    if P.repeats:
        other = family_lib(P.n_other, P.other_median, "ncRNA")
        mrna = isoform_lib(P.n_mrna, P.mrna_median, P.mrna_cap, "NM")
    else:
        other = lognormal_lib(P.n_other, P.other_median, 5000, 50, "ncRNA")
        mrna = lognormal_lib(P.n_mrna, P.mrna_median, P.mrna_cap, 200, "NM")
    return {"mirna": mirna, "hairpin": hairpin, "other": other, "mrna": mrna}, merges


@dataclasses.dataclass
class Mixture:
    mirna: float = 0.45
    hairpin: float = 0.05
    other: float = 0.25
    mrna: float = 0.10
    random: float = 0.15
    # inside the mirna class
    m_exact: float = 0.70
    m_templ: float = 0.15
    m_nontempl: float = 0.10
    m_subst: float = 0.05
    n_frac: float = 0.002
    hairpin_mm_max: int = 1
    # Recurring fragments (real small-RNA libraries are dominated by a limited set of tRNA / rRNA fragments and of
    # unannotated sequences: docs/miRge/miRge.exampleOutput/report.html has 9 % of the other-ncRNA reads and 25 % of the
    # remaining reads distinct).  pool_* = size of a fixed pool the class draws from (Zipf(1.0) weights; 0 = every
    # fragment fresh), fresh_* = fraction of the class drawn fresh all the same.
    pool_other: int = 0
    fresh_other: float = 1.0
    pool_random: int = 0
    fresh_random: float = 1.0

    @staticmethod
    def default() -> "Mixture":
        return Mixture()

    @staticmethod
    def realistic() -> "Mixture":
        """Class proportions and redundancy of the reference's only published run (report.html:159-175: 1.44M trimmed reads,
        188,088 distinct; reads 36 % miRNA, 0.07 % hairpin, 39 % other ncRNA, 3.7 % mRNA, 21 % remaining): U/N ~ 0.13 at 20M reads."""
        return Mixture(mirna=0.36, hairpin=0.001, other=0.39, mrna=0.037, random=0.212,
                       m_exact=0.85, m_templ=0.10, m_nontempl=0.04, m_subst=0.01,
                       pool_other=400_000, fresh_other=0.05, pool_random=800_000, fresh_random=0.06)

    @staticmethod
    def isomir_rich() -> "Mixture":      # BASELINE.json configs[3]
        return Mixture(mirna=0.70, hairpin=0.15, other=0.0, mrna=0.0, random=0.15,
                       m_exact=0.40, m_templ=0.25, m_nontempl=0.10, m_subst=0.25, hairpin_mm_max=2)


def _gather(text, start, lens, W):
    idx = start[:, None] + np.arange(W, dtype=np.int64)[None, :]
    np.clip(idx, 0, len(text) - 1, out=idx)
    out = text[idx]
    out[np.arange(W)[None, :] >= lens[:, None]] = 0
    return out


def _substitute(rng, seqs, lens, nsub):
    """Replace nsub[i] distinct positions of read i by a different base."""
    n = len(lens)
    for k in range(int(nsub.max()) if n else 0):
        rows = np.nonzero(nsub > k)[0]
        if not len(rows):
            break
        # distinct positions: stratify the read into thirds by k
        seg = lens[rows] // 3
        pos = k * seg + rng.integers(0, np.maximum(seg, 1))
        cur = seqs[rows, pos]
        code = (cur == ord("C")) * 1 + (cur == ord("G")) * 2 + (cur == ord("T")) * 3
        seqs[rows, pos] = ACGT[(code + rng.integers(1, 4, size=len(rows))) % 4]


def _fragments(rng, lib: Library, n, lo, hi, W):
    """n fragments with length U[lo,hi], start uniform over the library text, kept inside one reference."""
    if n == 0:
        return np.zeros((0, W), np.uint8), np.zeros(0, np.int64)
    g = rng.integers(0, len(lib.text), size=n)
    ref = np.searchsorted(lib.offs, g, side="right") - 1
    rl = lib.offs[ref + 1] - lib.offs[ref]
    lens = np.minimum(rng.integers(lo, hi + 1, size=n), rl)
    start = np.minimum(g, lib.offs[ref + 1] - lens)
    return _gather(lib.text, start, lens, W), lens


_POOLS = {}


def _pooled(rng, key, make_pool, n, fresh_frac, make_fresh, W):
    """n reads of one class: a fraction fresh_frac made by make_fresh(rng, m), the others drawn with Zipf(1.0) weights
    from a fixed pool (built once per process by make_pool from a seed of its own, so every chunk and every rank of a
    run draws from the same pool)."""
    if key not in _POOLS:
        ps, pl = make_pool()
        w = 1.0 / np.arange(1, len(pl) + 1, dtype=np.float64)
        _POOLS[key] = (ps, pl, np.cumsum(w / w.sum()))
    ps, pl, cdf = _POOLS[key]
    nf = int(round(n * fresh_frac))
    fs, fl = make_fresh(rng, nf)
    pick = np.minimum(np.searchsorted(cdf, rng.random(n - nf)), len(pl) - 1)
    return np.concatenate([fs, ps[pick]]), np.concatenate([fl, pl[pick]])


def make_reads(libs, n: int, seed: int = BASE_SEED + 1, mixture: Mixture | None = None,
               W: int = 48, zipf_perm_seed: int | None = None):
    """n adapter-trimmed reads: (seqs uint8[n, W] zero padded ASCII, lens uint8[n]).  Lengths 16..47."""
    mx = mixture or Mixture.default()
    rng = np.random.default_rng(seed)
    fr = np.array([mx.mirna, mx.hairpin, mx.other, mx.mrna, mx.random], dtype=np.float64)
    cnt = np.floor(fr / fr.sum() * n).astype(np.int64); cnt[0] += n - cnt.sum()
    parts_s, parts_l = [], []
    # ---- miRNA-derived, Zipf(1.1) abundance over a permutation of the entries
    mir = libs["mirna"]
    prng = np.random.default_rng(zipf_perm_seed if zipf_perm_seed is not None else seed ^ 0x5A17F)
    w = 1.0 / np.arange(1, mir.n + 1, dtype=np.float64) ** 1.1
    w = w[prng.permutation(mir.n)]; w /= w.sum()
    nm = int(cnt[0])
    ent = rng.choice(mir.n, size=nm, p=w)
    el = (mir.offs[ent + 1] - mir.offs[ent])
    kind = rng.choice(4, size=nm, p=np.array([mx.m_exact, mx.m_templ, mx.m_nontempl, mx.m_subst]) /
                      (mx.m_exact + mx.m_templ + mx.m_nontempl + mx.m_subst))
    mat = np.minimum(el, 22 + (ent % 3))                    # canonical mature length 22..24, fixed per entry
    a = np.zeros(nm, np.int64); ln = mat.copy()
    t = kind == 1                                            # templated 5'/3' variants inside the entry
    a[t] = rng.integers(0, 3, size=int(t.sum()))
    ln[t] = np.clip(mat[t] + rng.integers(-2, 4, size=int(t.sum())), 16, None)
    ln = np.minimum(ln, el - a)
    s = _gather(mir.text, mir.offs[ent] + a, ln, W)
    nt = np.nonzero(kind == 2)[0]                            # 3' non-templated A/T additions of 1-2 nt
    add = rng.integers(1, 3, size=len(nt))
    for k in range(2):
        rows = nt[add > k]
        s[rows, ln[rows]] = np.where(rng.random(len(rows)) < 0.5, ord("A"), ord("T")).astype(np.uint8)
        ln[rows] += 1
    nsub = np.zeros(nm, np.int64); sb = kind == 3
    nsub[sb] = rng.integers(1, 3, size=int(sb.sum()))
    _substitute(rng, s, ln, nsub)
    parts_s.append(s); parts_l.append(ln)
    # ---- hairpin fragments 26-40 nt, 0..hairpin_mm_max substitutions
    s, ln = _fragments(rng, libs["hairpin"], int(cnt[1]), 26, 40, W)
    _substitute(rng, s, ln, rng.integers(0, mx.hairpin_mm_max + 1, size=len(ln)))
    parts_s.append(s); parts_l.append(ln)
    # ---- other ncRNA fragments 16-40 nt, 0-1 substitutions
    def other_frags(r, m):
        fs, fl = _fragments(r, libs["other"], m, 16, 40, W)
        _substitute(r, fs, fl, r.integers(0, 2, size=len(fl)))
        return fs, fl
    if mx.pool_other:
        s, ln = _pooled(rng, ("other", id(libs["other"].text), mx.pool_other, W),
                        lambda: other_frags(np.random.default_rng(BASE_SEED ^ 0x07E2), mx.pool_other), int(cnt[2]), mx.fresh_other, other_frags, W)
    else:
        s, ln = other_frags(rng, int(cnt[2]))
    parts_s.append(s); parts_l.append(ln)
    # ---- mRNA fragments 16-45 nt, exact
    s, ln = _fragments(rng, libs["mrna"], int(cnt[3]), 16, 45, W)
    parts_s.append(s); parts_l.append(ln)
    # ---- uniform random 16-45 nt
    def random_reads(r, m):
        fl = r.integers(16, 46, size=m)
        fs = ACGT[r.integers(0, 4, size=(m, W))]
        fs[np.arange(W)[None, :] >= fl[:, None]] = 0
        return fs, fl
    if mx.pool_random:
        s, ln = _pooled(rng, ("random", mx.pool_random, W), lambda: random_reads(np.random.default_rng(BASE_SEED ^ 0x4A4D), mx.pool_random),
                        int(cnt[4]), mx.fresh_random, random_reads, W)
    else:
        s, ln = random_reads(rng, int(cnt[4]))
    parts_s.append(s); parts_l.append(ln)
    seqs = np.concatenate(parts_s); lens = np.concatenate(parts_l)
    # ---- 0.2 % of reads get one N
    rows = np.nonzero(rng.random(n) < mx.n_frac)[0]
    seqs[rows, rng.integers(0, lens[rows])] = ord("N")
    perm = rng.permutation(n)
    seqs = np.ascontiguousarray(seqs[perm]); lens = lens[perm]
    assert lens.min() >= 1 and lens.max() < W
    return seqs, lens.astype(np.uint8)


def reads_to_list(seqs, lens):
    return [seqs[i, :lens[i]].tobytes().decode() for i in range(len(lens))]


def write_fastq(path, seqs, lens, name_prefix="r"):
    """4-line FASTQ records as reference miRge.pl:300-303 reads them; qualities all 'I'."""
    with open(path, "w") as f:
        for i in range(len(lens)):
            s = seqs[i, :lens[i]].tobytes().decode()
            f.write("@%s%d\n%s\n+\n%s\n" % (name_prefix, i, s, "I" * len(s)))


def write_fastq_fast(path, seqs, lens, block=1 << 20):
    """The same records as write_fastq (names "@r%09d"), assembled with numpy: for the multi-million-read files of
    bench.py's end-to-end leg.  Returns the number of bytes written."""
    n, W = seqs.shape
    total = 0
    u8 = np.uint8
    with open(path, "wb") as f:
        for b in range(0, n, block):
            s = seqs[b:b + block]; l = lens[b:b + block].astype(np.int64); m = len(l)
            idx = np.arange(b, b + m, dtype=np.int64)
            digits = ((idx[:, None] // 10 ** np.arange(8, -1, -1, dtype=np.int64)[None, :]) % 10 + 48).astype(u8)
            valid = np.arange(W)[None, :] < l[:, None]
            one = np.ones((m, 1), dtype=bool)
            row = np.concatenate([np.full((m, 1), ord("@"), u8), np.full((m, 1), ord("r"), u8), digits, np.full((m, 1), 10, u8),
                                  s, np.tile(np.frombuffer(b"\n+\n", dtype=u8), (m, 1)), np.full((m, W), ord("I"), u8),
                                  np.full((m, 1), 10, u8)], axis=1)
            mask = np.concatenate([np.repeat(one, 12, axis=1), valid, np.repeat(one, 3, axis=1), valid, one], axis=1)
            out = row[mask]
            f.write(out.tobytes()); total += out.size
    return total


def write_species_dir(path, libs, merges, with_dummy_ebwt=True):
    """miRge.seqLibs/<species>/ layout (reference miRge.pl:92-97, 189-192, 502): the four
    index basenames plus merges.csv.  Sequences travel as <lib>.fa beside the (dummy) .ebwt."""
    import os
    os.makedirs(path, exist_ok=True)
    for nm in LIB_NAMES:
        libs[nm].write_fasta(os.path.join(path, nm + ".fa"))
        if with_dummy_ebwt:
            open(os.path.join(path, nm + ".1.ebwt"), "wb").close()
    with open(os.path.join(path, "merges.csv"), "w") as f:
        for row in merges:
            f.write(",".join(row) + "\n")


ILLUMINA_ADAPTER = "TGGAATTCTCGGGTGCCAAGGAACTCCAG"      # reference miRge.pl:82


def make_raw_fastq(seqs, lens, seed: int, adapter: str = ILLUMINA_ADAPTER, read_len: int = 50, phred: int = 33,
                   adapter_error: float = 0.15, low_tail: float = 0.25, crlf: bool = False, odd: float = 0.0) -> bytes:
    """Raw reads as a sequencer writes them: insert + 3' adapter + random fill, cut to read_len.  A share
    of the reads carries one or two errors (substitution / insertion / deletion) in the adapter part, a share a
    low-quality 3' tail, a few a low-quality base in the middle or an N; a share `odd` is lower case, mixed case, or holds an IUPAC code or '.' (reads
    no key holds).  Returns FASTQ text."""
    rng = np.random.default_rng(seed)
    acgt = "ACGT"
    hi, lo = chr(phred + 40), chr(phred + 2)
    nl = "\r\n" if crlf else "\n"
    out = []
    for i in range(len(lens)):
        ins = seqs[i, :lens[i]].tobytes().decode()
        ad = list(adapter)
        if rng.random() < adapter_error:
            for _ in range(int(rng.integers(1, 3))):
                p = int(rng.integers(0, len(ad))); kind = rng.random()
                if kind < 0.6:
                    ad[p] = acgt[(acgt.index(ad[p]) + int(rng.integers(1, 4))) % 4]
                elif kind < 0.8:
                    ad.insert(p, acgt[int(rng.integers(0, 4))])
                else:
                    del ad[p]
        r = ins + "".join(ad) + "".join(acgt[int(x)] for x in rng.integers(0, 4, size=read_len))
        r = r[:read_len]
        q = [hi] * len(r)
        if rng.random() < low_tail:
            t = int(rng.integers(1, 20))
            for p in range(max(0, len(r) - t), len(r)):
                q[p] = lo if rng.random() < 0.85 else hi
        if rng.random() < 0.05:
            q[int(rng.integers(0, len(r)))] = lo
        if rng.random() < 0.02:
            p = int(rng.integers(0, len(r))); r = r[:p] + "N" + r[p + 1:]
        if odd and rng.random() < odd:
            kind = rng.random()
            if kind < 0.3:
                r = r.lower()
            elif kind < 0.5:
                r = "".join(c.lower() if rng.random() < 0.5 else c for c in r)
            else:
                p = int(rng.integers(0, len(r))); r = r[:p] + str(rng.choice(list("RYKMSWBDHV.x"))) + r[p + 1:]
        out.append("@raw%d%s%s%s+%s%s%s" % (i, nl, r, nl, nl, "".join(q), nl))
    return "".join(out).encode()


def write_raw_fastq_fast(path, seqs, lens, seed: int, adapter: str = ILLUMINA_ADAPTER, read_len: int = 50, block=1 << 20):
    """Raw fixed-length reads (insert + adapter + random fill, cut to read_len; a quarter with a low-quality 3' tail),
    assembled with numpy for bench.py's multi-million-read raw leg.  Returns the number of bytes written."""
    rng = np.random.default_rng(seed)
    n, W = seqs.shape
    u8 = np.uint8
    ad = np.frombuffer(adapter.encode(), dtype=u8)
    acgt = np.frombuffer(b"ACGT", dtype=u8)
    total = 0
    with open(path, "wb") as f:
        for b in range(0, n, block):
            sq = seqs[b:b + block]; l = lens[b:b + block].astype(np.int64); m = len(l)
            j = np.arange(read_len, dtype=np.int64)[None, :]
            rel = j - l[:, None]
            row = acgt[rng.integers(0, 4, size=(m, read_len))]
            ins = rel < 0
            row[ins] = np.take_along_axis(np.pad(sq, ((0, 0), (0, max(0, read_len - W)))), np.broadcast_to(j, (m, read_len)), axis=1)[ins]
            ina = (rel >= 0) & (rel < len(ad))
            row[ina] = ad[np.clip(rel, 0, len(ad) - 1)][ina]
            qual = np.full((m, read_len), ord("I"), u8)
            tail = rng.integers(1, 20, size=m) * (rng.random(m) < 0.25)
            qual[j >= (read_len - tail)[:, None]] = ord("#")
            idx = np.arange(b, b + m, dtype=np.int64)
            digits = ((idx[:, None] // 10 ** np.arange(8, -1, -1, dtype=np.int64)[None, :]) % 10 + 48).astype(u8)
            rec = np.concatenate([np.full((m, 1), ord("@"), u8), np.full((m, 1), ord("r"), u8), digits, np.full((m, 1), 10, u8),
                                  row, np.tile(np.frombuffer(b"\n+\n", dtype=u8), (m, 1)), qual, np.full((m, 1), 10, u8)], axis=1)
            f.write(rec.tobytes()); total += rec.size
    return total
