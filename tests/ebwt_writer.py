"""Writes <base>.1.ebwt / .3.ebwt / .4.ebwt for a set of references in the layout mirge_amd/csrc/ebwt.h describes
(bowtie 1 small index, restated from memory: UNVALIDATED against a real bowtie-build, which is not available).
Only the parts the reader uses carry meaning; the FM-index sections of .1.ebwt are filled with sized dummy bytes."""
import struct


def write_ebwt(base, names, seqs, big_endian=False, line_rate=6, lines_per_side=1, off_rate=5, ftab_chars=4, entire_reverse=False):
    e = ">" if big_endian else "<"
    recs, bases = [], []
    kept_names = []
    for nm, s in zip(names, seqs):
        first, gap, i = True, 0, 0
        s = s.upper()
        any_rec = False
        while i < len(s):
            if s[i] not in "ACGT":
                gap += 1; i += 1
                continue
            j = i
            while j < len(s) and s[j] in "ACGT":
                j += 1
            recs.append((gap, j - i, 1 if first else 0))
            bases.append(s[i:j])
            first, gap, i, any_rec = False, 0, j, True
        if any_rec:
            kept_names.append(nm)                 # a reference with no unambiguous base is skipped by bowtie-build, name included
    allb = "".join(bases)
    with open(base + ".3.ebwt", "wb") as f:
        f.write(struct.pack(e + "iI", 1, len(recs)))
        for off, ln, first in recs:
            f.write(struct.pack(e + "IIB", off, ln, first))
    packed = bytearray((len(allb) + 3) // 4)
    for i, ch in enumerate(allb):
        packed[i >> 2] |= "ACGT".index(ch) << (2 * (i & 3))
    with open(base + ".4.ebwt", "wb") as f:
        f.write(bytes(packed))
    ln = len(allb)
    side_sz = lines_per_side << line_rate
    side_bwt_sz = side_sz - 8
    bwt_sz = ln // 4 + 1
    ebwt_tot = (bwt_sz + 2 * side_bwt_sz - 1) // (2 * side_bwt_sz) * 2 * side_sz
    flags = -4 if entire_reverse else 1
    with open(base + ".1.ebwt", "wb") as f:
        f.write(struct.pack(e + "iIiiiii", 1, ln, line_rate, lines_per_side, off_rate, ftab_chars, flags))
        f.write(struct.pack(e + "I", len(kept_names)))
        for nm, s in zip(names, seqs):
            if nm in kept_names:
                f.write(struct.pack(e + "I", sum(1 for ch in s.upper() if ch in "ACGT")))
        f.write(struct.pack(e + "I", len(recs)))
        f.write(b"\x07" * (12 * len(recs)))
        f.write(b"\xA5" * ebwt_tot)
        f.write(struct.pack(e + "I", 0))
        f.write(b"\x00" * 20)
        f.write(b"\x11" * (((1 << (2 * ftab_chars)) + 1) * 4))
        f.write(b"\x00" * (ftab_chars * 2 * 4))
        f.write(("\n".join(kept_names) + "\n").encode() + b"\0")
    # what a reader must give back: trailing ambiguous characters are not recorded
    want = []
    for nm, s in zip(names, seqs):
        if nm in kept_names:
            t = "".join(ch if ch in "ACGT" else "N" for ch in s.upper())
            want.append((nm, t.rstrip("N")))
    return want
