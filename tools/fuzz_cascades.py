#!/usr/bin/env python3
"""Randomised cascades on the GPU against the CPU oracle, many seeds (run on the GPU box):
   python tools/fuzz_cascades.py [first_seed] [n_seeds]
Each seed runs the randomised parity tests of tests/ -- generic kernel only, every library with slot tables (k-mer
length swept; probe kernel and LDS path) and cascades that mix both kinds of round -- with that seed.  Prints one line per mismatch."""
import os
import sys

sys.path.insert(0, os.getcwd())
import pytest  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.lib()
from tests import test_gpu_parity as tp, test_gpu_probe as tq  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    mp = pytest.MonkeyPatch()
    bad = 0
    for seed in range(first, first + count):
        cases = [("generic", lambda: tp.test_random_cascades_vs_oracle(orc, seed)),
                 ("slots kb=%d" % (6 + seed % 7), lambda: tq.test_random_cascades_through_slot_tables(orc, mp, seed, 6 + seed % 7, "0")),
                 ("slots kb=%d lds" % (6 + seed % 7), lambda: tq.test_random_cascades_through_slot_tables(orc, mp, seed, 6 + seed % 7, "1")),
                 ("mixed", lambda: tq.test_cascades_that_mix_probe_and_generic_rounds(orc, mp, seed, str(seed & 1)))]
        for name, fn in cases:
            try:
                fn()
            except AssertionError as e:
                msg = str(e)
                if not msg or "> 1000" in msg or "assert sum" in msg or "assert min(" in msg:     # the tests' own sanity floors (the only asserts without a message), not a mismatch
                    continue
                bad += 1
                print("seed", seed, name, "MISMATCH", msg[:400], flush=True)
            finally:
                mp.undo()
        if seed % 10 == 9:
            print("through seed", seed, flush=True)
    print("mismatching cascades:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
