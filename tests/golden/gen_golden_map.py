"""What the REFERENCE's get_read_seq / map_read / _io_and_map_read (tombo/resquiggle.py:1221-1465)
make of a FAST5 read and an aligner hit (build container only) -> tests/golden/map_cases.json.

mappy is not installed (and minimap2 is out of scope): the hits are scripted
(tests/scripted_aligner.py) and the FAST5 is the dict-backed stand-in of tests/memh5.py, so what
is recorded is the reference's handling of a hit -- clip counts, cigar accounting, the extended
reference window per strand and sample type, error strings, the index record and its filters.
The test (tests/test_mapping_glue.py) rebuilds the same inputs from `cases()` below and compares.
"""
import os
import sys
import json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import memh5  # noqa: E402
from scripted_aligner import ScriptedAligner, Hit  # noqa: E402


def make_fast5(read_seq, qual, n_raw, seed, read_id=b'read-%d', with_channel=True, fastq=True,
               bc_subgrp='BaseCalled_template'):
    """A single-read FAST5 tree: /Raw/Reads/Read_N/Signal (+ read_id), the Fastq slot, channel_id"""
    f = memh5.MemGroup()
    rng = np.random.RandomState(seed)
    rd = f.create_group('Raw/Reads/Read_%d' % seed)
    rd.create_dataset('Signal', data=rng.randint(300, 900, n_raw).astype(np.int16))
    if read_id is not None:
        rd.attrs['read_id'] = read_id % seed if b'%d' in read_id else read_id
    rd.attrs['read_num'] = seed
    if fastq:
        g = f.create_group('Analyses/Basecall_1D_000/' + bc_subgrp)
        g.create_dataset('Fastq', data=np.bytes_(('@r%d\n%s\n+\n%s\n' % (seed, read_seq, qual)).encode()))
    if with_channel:
        c = f.create_group('UniqueGlobalKey/channel_id')
        c.attrs.update(dict(offset=np.float64(10.0), range=np.float64(1400.5),
                            digitisation=np.float64(8192.0), channel_number=b'17',
                            sampling_rate=np.float64(4000.0)))
    return f


def mutate(rng, s):
    """a read of reference stretch s with a few substitutions"""
    s = list(s)
    for i in rng.choice(len(s), max(1, len(s) // 25), replace=False):
        s[i] = 'ACGT'[('ACGT'.index(s[i]) + 1 + rng.randint(3)) % 4]
    return ''.join(s)


def cases():
    """name -> dict(sample type, fast5 pieces, hits, kwargs); deterministic"""
    rng = np.random.RandomState(7)
    chrA = ''.join(rng.choice(list('ACGT'), 600))
    chrB = ''.join(rng.choice(list('ACGT'), 300))
    records = {'chrA': chrA, 'chrB': chrB}
    comp = str.maketrans('ACGT', 'TGCA')
    out = {}

    def add(name, samp, ctg, r_st, r_en, strand, lead, tail, cigar_extra, qual_char='5', **kw):
        ref = records[ctg][r_st:r_en]
        body = mutate(rng, ref if strand == 1 else ref.translate(comp)[::-1])
        lead_s = ''.join(rng.choice(list('ACGT'), lead))
        tail_s = ''.join(rng.choice(list('ACGT'), tail))
        read = lead_s + body + tail_s
        cigar = [[r_en - r_st - sum(l for l, op in cigar_extra if op in (0, 2, 3, 7, 8)), 0]] + \
            [list(c) for c in cigar_extra]
        fq_seq = read.replace('T', 'U') if samp == 'RNA' else read
        hit = dict(ctg=ctg, r_st=r_st, r_en=r_en, strand=strand, mlen=r_en - r_st - 11,
                   cigar=cigar, q_st=lead, q_en=lead + len(body))
        out[name] = dict(samp=samp, read=fq_seq, qual=qual_char * len(read), n_raw=1500 + 10 * len(read),
                         seed=100 + len(out), hits=[hit], kw=kw)

    add('dna_plus', 'DNA', 'chrA', 50, 250, 1, 7, 4, [(3, 1), (5, 2), (20, 7), (2, 8), (4, 3), (6, 6)])
    add('dna_minus', 'DNA', 'chrA', 300, 520, -1, 0, 9, [(2, 1), (1, 2)])
    add('rna_plus', 'RNA', 'chrB', 40, 200, 1, 3, 0, [(4, 2)])
    add('rna_minus', 'RNA', 'chrB', 60, 260, -1, 5, 6, [(1, 1)])
    add('dna_plus_at_record_start', 'DNA', 'chrB', 0, 120, 1, 2, 2, [])
    add('dna_minus_at_record_start', 'DNA', 'chrB', 1, 150, -1, 2, 2, [])
    add('rna_minus_to_record_end', 'RNA', 'chrB', 150, 300, -1, 0, 0, [])
    add('dna_two_hits_first_wins', 'DNA', 'chrA', 100, 220, 1, 1, 1, [])
    out['dna_two_hits_first_wins']['hits'].append(dict(ctg='chrB', r_st=5, r_en=100, strand=-1, mlen=60,
                                                       cigar=[[95, 0]], q_st=0, q_en=95))
    add('dna_unicode_fastq_low_q', 'DNA', 'chrA', 10, 90, 1, 0, 0, [], qual_char='#')
    # failures
    add('err_no_hit', 'DNA', 'chrA', 10, 90, 1, 0, 0, [])
    out['err_no_hit']['hits'] = []
    add('err_bad_cigar', 'DNA', 'chrA', 10, 90, 1, 0, 0, [(3, 4)])
    add('err_len_range', 'DNA', 'chrA', 10, 90, 1, 0, 0, [], seq_len_rng=[100, 1000])
    add('ok_len_range', 'DNA', 'chrA', 10, 190, 1, 0, 0, [], seq_len_rng=[100, 1000])
    add('err_q_score', 'DNA', 'chrA', 10, 90, 1, 0, 0, [], qual_char='$', q_score_thresh=7.0)
    add('err_no_fastq', 'DNA', 'chrA', 10, 90, 1, 0, 0, [])
    out['err_no_fastq']['no_fastq'] = True
    add('err_unknown_contig', 'DNA', 'chrA', 10, 90, 1, 0, 0, [])
    out['err_unknown_contig']['hits'][0]['ctg'] = 'chrZ'
    add('err_signal_len', 'DNA', 'chrA', 10, 90, 1, 0, 0, [], sig_len_rng=[10, 500])
    add('err_ref_has_n', 'DNA', 'chrA', 10, 90, 1, 0, 0, [])
    out['err_ref_has_n']['patch_ref'] = ['chrA', 40, 'N']
    return records, out


def build(case, records):
    rec = dict(records)
    if 'patch_ref' in case:
        c, i, ch = case['patch_ref']
        rec[c] = rec[c][:i] + ch + rec[c][i + 1:]
    f = make_fast5(case['read'], case['qual'], case['n_raw'], case['seed'],
                   fastq=not case.get('no_fastq'))
    hits = [Hit(h['ctg'], h['r_st'], h['r_en'], h['strand'], h['mlen'],
                [tuple(c) for c in h['cigar']], h['q_st'], h['q_en']) for h in case['hits']]
    key = case['read'].replace('U', 'T') if case['samp'] == 'RNA' else case['read']
    return f, ScriptedAligner(rec, {key: hits})


class _Q(object):
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)


class _Conn(object):
    """the pipe to the resquiggle worker: answers every mapped read with a scripted result"""
    def __init__(self, th, score, segs_step):
        self.th, self.score, self.segs_step, self.sent = th, score, segs_step, []

    def send(self, x):
        self.sent.append(x)

    def recv(self):
        mr = self.sent[-1][0]
        nb = len(mr.genome_seq) - 5
        segs = np.arange(nb + 1, dtype=np.int64) * self.segs_step
        res = mr._replace(read_start_rel_to_raw=123, segs=segs, sig_match_score=self.score,
                          scale_values=self.th.scaleValues(1.0, 2.0, -5.0, 5.0, 5.0),
                          raw_signal=np.zeros(int(segs[-1])), norm_params_changed=False)
        return False, res


def main():
    import ref_oracle
    rq, ts, th = ref_oracle.load()
    from tombo_amd import tombo_stats as my_ts, tombo_helper as my_th
    rq._DRY_RUN = True                    # no FAST5 writer in this generator (gen_golden_fast5.py)
    records, cs = cases()
    out = {}
    for name, case in cs.items():
        samp = th.seqSampleType(case['samp'], case['samp'] == 'RNA')
        mm = my_ts.TomboModel(seq_samp_type=my_th.seqSampleType(case['samp'], case['samp'] == 'RNA'))

        class StdRef(object):
            kmer_width, central_pos = mm.kmer_width, mm.central_pos
        kw = dict(case['kw'])
        sig_len_rng = kw.pop('sig_len_rng', None)
        rec = {}
        # --- map_read
        f, al = build(case, records)
        try:
            mr = rq.map_read(f, al, StdRef, samp, q_score_thresh=kw.get('q_score_thresh', 0),
                             seq_len_rng=kw.get('seq_len_rng'))
            rec['map_read'] = dict(align_info=list(mr.align_info), genome_loc=list(mr.genome_loc),
                                   genome_seq=mr.genome_seq, mean_q_score=float(mr.mean_q_score),
                                   start_clip_bases=mr.start_clip_bases, drained=al.n_drained)
        except th.TomboError as e:
            rec['map_read'] = dict(error=str(e))
        except Exception as e:      # the reference's own slips (raise th.TomboReads(...)) land here
            rec['map_read'] = dict(unexpected=type(e).__name__)
        # --- _io_and_map_read: index record and filters (two scores, with / without obs filter)
        rec['io'] = []
        for score, obs_filter, step in ((0.9, None, 7), (1.3, None, 7), (0.9, [(99, 5)], 7),
                                        (0.9, [(99, 50)], 7)):
            f, al = build(case, records)
            failed, index = _Q(), _Q()
            raised = None
            try:
                rq._io_and_map_read(
                    f, failed, ['BaseCalled_template'], 'Basecall_1D_000', 'RawGenomeCorrected_000', al,
                    samp, None, 'dir/read_%d.fast5' % case['seed'], 0, _Conn(th, score, step), 5.0, True,
                    obs_filter, index, kw.get('q_score_thresh', 0), 1.1, StdRef, sig_len_rng,
                    kw.get('seq_len_rng'))
            except th.TomboError as e:   # raised before the per-subgroup handler (signal checks)
                raised = str(e)
            idx = [[c, s] + [x.item() if hasattr(x, 'item') else x for x in rd] for c, s, rd in index.items]
            rec['io'].append(dict(score=score, obs_filter=obs_filter, step=step, index=idx, raised=raised,
                                  failed=[[m if t else 'unexpected', t] for m, _, t in failed.items]))
        out[name] = rec
    path = os.path.join(HERE, 'map_cases.json')
    with open(path, 'w') as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
    print('map_cases.json %.1f KB, %d cases' % (os.path.getsize(path) / 1024., len(out)))
    for k, v in out.items():
        print(k, {a: (b if a != 'genome_seq' else b[:12] + '...') for a, b in v['map_read'].items()},
              [(x['failed'], len(x['index'])) for x in v['io']][:2])


if __name__ == '__main__':
    main()
