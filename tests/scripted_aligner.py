"""A stand-in for `mappy.Aligner` in the tests of the mapping glue (tombo_amd/mapping.py): hits are
scripted per read sequence instead of computed -- the aligner (minimap2) is third party and out
of scope, what is under test is everything the reference does with a hit.  The same object is
handed to the reference's `map_read` by tests/golden/gen_golden_map.py."""
from collections import namedtuple

Hit = namedtuple('Hit', ('ctg', 'r_st', 'r_en', 'strand', 'mlen', 'cigar', 'q_st', 'q_en'))


class ScriptedAligner(object):
    def __init__(self, records, hits):
        self.records = dict(records)   # contig name -> sequence
        self.hits = dict(hits)         # read sequence -> [Hit, ...]
        self.n_drained = 0

    def map(self, seq, buf=None):
        for h in self.hits.get(seq, []):
            yield h
        self.n_drained += 1            # the caller has to exhaust the iterator (mappy leak)

    def seq(self, ctg, start, end):
        rec = self.records.get(ctg)
        if rec is None or start < 0 or start >= len(rec):
            return None
        return rec[start:end]
