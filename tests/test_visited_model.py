"""A sequential model of the search kernels' visited set (flatnav_amd/csrc/visited.hpp) and the argument DESIGN.md makes
for it: three places -- a two-choice table of 16-bit tags in buckets of four (no relocation), a 64-word stash of full ids in
16 hashed buckets of four, a bitmap -- where an id goes to the first place that has room and nothing is ever removed.  The
claim: "both table buckets full -> ask the stash" and "stash bucket full -> ask the bitmap" stay true once they are true,
so test-and-mark behaves like a plain set whatever the table size.  The HIP code itself is checked against the oracle on
the GPU (tests/test_gpu_parity.py: spill paths, wide tags); this checks the scheme on the CPU, with the kernels' hash
constants.  Replaces nothing in the reference: include/flatnav/util/VisitedSetPool.h:16-50 is a plain mark array."""
import random

import pytest

M32 = 0xFFFFFFFF


class ModelVisited:
    def __init__(self, slots, nbits, stash_words=64):
        assert slots % 4 == 0 and (slots // 4) & (slots // 4 - 1) == 0
        self.k = (slots // 4).bit_length() - 1
        self.nbits, self.t = nbits, nbits - self.k
        assert 0 <= self.t <= 14  # 16-bit tags: (remainder << 1 | which) + 1
        self.table = [[0, 0, 0, 0] for _ in range(slots // 4)]
        self.stash = [[0, 0, 0, 0] for _ in range(stash_words // 4)]
        self.stash_shift = 32 - ((stash_words // 4).bit_length() - 1)
        self.bitmap = set()
        self.where = {}

    def _slot(self, ident, mult, which):
        h = (ident * mult) & ((1 << self.nbits) - 1)  # a bijection on nbits-bit integers (odd multiplier)
        return h >> self.t, (((h & ((1 << self.t) - 1)) << 1) + 1 + which)

    def insert(self, ident):
        """test-and-mark: True when the id was new"""
        (b1, t1), (b2, t2) = self._slot(ident, 0x9E3779B1, 0), self._slot(ident, 0x85EBCA6B, 1)
        B1, B2 = self.table[b1], self.table[b2]
        if t1 in B1 or t2 in B2:
            return False
        f1, f2 = 4 - B1.count(0), 4 - B2.count(0)
        if min(f1, f2) < 4:  # the emptier bucket, its lowest free field
            (B1 if f1 <= f2 else B2)[min(f1, f2)] = t1 if f1 <= f2 else t2
            self.where[ident] = "table"
            return True
        S = self.stash[((ident * 0xC2B2AE35) & M32) >> self.stash_shift]
        if ident + 1 in S:
            return False
        if 0 in S:
            S[S.index(0)] = ident + 1
            self.where[ident] = "stash"
            return True
        if ident in self.bitmap:
            return False
        self.bitmap.add(ident)
        self.where[ident] = "bitmap"
        return True


@pytest.mark.parametrize("slots,nbits,n_ids", [(256, 20, 600), (256, 17, 3000), (1024, 20, 900), (2048, 21, 2500), (4096, 24, 2000)])
def test_three_level_visited_set_is_a_set(slots, nbits, n_ids):
    for seed in range(20):
        rng = random.Random(1000 * slots + seed)
        vs, seen = ModelVisited(slots, nbits), set()
        ids = [rng.randrange(1 << nbits) for _ in range(n_ids)]
        stream = ids + [rng.choice(ids) for _ in range(2 * n_ids)]  # every id again, twice on average, in random order
        rng.shuffle(stream)
        for ident in stream:
            assert vs.insert(ident) == (ident not in seen), (slots, seed, ident, vs.where.get(ident))
            seen.add(ident)
        # every id sits in exactly one place, and small tables did use all three
        assert len(vs.where) == len(seen)
        if n_ids > 2 * slots:
            assert {"table", "stash", "bitmap"} == set(vs.where.values())


def test_tags_identify_ids_exactly():
    # bucket index + remainder + hash selector reconstruct the id: two different ids never share (bucket, tag)
    vs = ModelVisited(256, 16)
    for mult, which in ((0x9E3779B1, 0), (0x85EBCA6B, 1)):
        assert len({vs._slot(i, mult, which) for i in range(1 << 16)}) == 1 << 16
