"""Golden vectors of the per-read statistics (row N4) from the REFERENCE (build container only).

    python tests/golden/gen_golden_stats.py        # writes tests/golden/stats_reads.npz

Runs the live reference's compute_de_novo_read_stats / compute_sample_compare_read_stats /
compute_alt_model_read_stats (tombo/tombo_stats.py:3675-4083) on synthetic resquiggled reads.
The reference loads `norm_mean` / `base` from a FAST5 file; here its three file accessors
(`h5py.File`, `th.get_multiple_slots_read_centric`, `th.get_single_slot_read_centric`,
`th.get_raw_read_slot`) are pointed at in-memory arrays -- everything after the file access is
the reference's own code (numpy, scipy.stats, the compiled c_calc_*llh* functions).  Only data is
written: the inputs (means, sequences, coordinates, control levels, the synthetic alt model) and
the outputs.
"""
import os
import sys
import json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import ref_oracle  # noqa: E402
from tombo_amd import tombo_stats as my_ts, tombo_helper as my_th  # noqa: E402

rq, ts, th = ref_oracle.load()

STORE = {}


class FakeFile(object):
    def __init__(self, fn, mode='r'):
        self.fn = fn

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Slot(object):
    def __init__(self, rid):
        self.attrs = {'read_id': rid}


def install():
    ts.h5py.File = FakeFile
    th.get_multiple_slots_read_centric = lambda f, names, grp=None: [STORE[f.fn][n] for n in names]
    th.get_single_slot_read_centric = lambda f, name, grp=None: STORE[f.fn][name]
    th.get_raw_read_slot = lambda f: _Slot(STORE[f.fn]['read_id'])


def main():
    install()
    rng = np.random.default_rng(2024)
    samp = th.seqSampleType('DNA', False)
    my_model = my_ts.TomboModel(seq_samp_type=my_th.seqSampleType('DNA', False))
    kmers = sorted(my_model.means.keys())
    std_ref = ts.TomboModel(kmer_ref=[(k, my_model.means[k], my_model.sds[k]) for k in kmers],
                            central_pos=my_model.central_pos, seq_samp_type=samp)
    K = std_ref.kmer_width
    # synthetic alternate model: 5mC-like, motif CG with the C modified; every k-mer with a C at
    # `pos` gets a shifted level
    alt_ref_rows = []
    for k in kmers:
        for pos in range(K):
            if k[pos] == 'C':
                alt_ref_rows.append((k, pos, my_model.means[k] + 0.35 * np.cos(hash_code(k, pos)),
                                     my_model.sds[k] * 1.1))
    alt_cg = ts.AltModel(kmer_ref=alt_ref_rows, central_pos=std_ref.central_pos, alt_base='C',
                         name='5mC_CG', motif=th.TomboMotif('CG', 1))
    alt_rows_a = [(k, pos, my_model.means[k] - 0.25 * np.sin(hash_code(k, pos)), my_model.sds[k])
                  for k in kmers for pos in range(K) if k[pos] == 'A']
    alt_a = ts.AltModel(kmer_ref=alt_rows_a, central_pos=std_ref.central_pos, alt_base='A',
                        name='6mA_GATC', motif=th.TomboMotif('GATC', 2))
    alt_refs = [('5mC_CG', alt_cg), ('6mA_GATC', alt_a)]

    out = {}
    cases = []
    specs = [dict(n=60, strand='+', start=1000), dict(n=180, strand='-', start=5000),
             dict(n=333, strand='+', start=200), dict(n=41, strand='-', start=77),
             dict(n=12, strand='+', start=10), dict(n=8, strand='-', start=3)]
    for ci, sp in enumerate(specs):
        n = sp['n']
        seq = ''.join('ACGT'[c] for c in rng.integers(0, 4, n))   # read-centric bases
        lev, _ = my_model.get_exp_levels_from_seq('AA' + seq + 'AAA')   # any plausible means
        means = lev[:n] + rng.normal(0, 0.3, n)
        fn = 'read_%d' % ci
        STORE[fn] = dict(norm_mean=means, base=np.frombuffer(seq.encode(), dtype='S1'),
                         read_id='rid_%d' % ci)
        r_data = th.readData(start=sp['start'], end=sp['start'] + n, filtered=False,
                             read_start_rel_to_raw=0, strand=sp['strand'], fn=fn,
                             corr_group='RawGenomeCorrected_000/BaseCalled_template', rna=False)
        out['c%d_means' % ci] = means
        out['c%d_seq' % ci] = np.array(seq)
        regs = [None, th.intervalData(chrm='c', start=sp['start'] + n // 4, end=sp['start'] + (3 * n) // 4,
                                      strand=sp['strand'])]
        for ri, reg in enumerate(regs):
            for fm in (0, 1, 2):
                tag = 'c%d_r%d_fm%d' % (ci, ri, fm)
                # de novo
                try:
                    pv, ps, rid = ts.compute_de_novo_read_stats(r_data, std_ref, fm, reg)
                    out[tag + '_dn_p'] = pv[ts.DE_NOVO_TXT]
                    out[tag + '_dn_pos'] = ps[ts.DE_NOVO_TXT]
                    err = ''
                except th.TomboError as e:
                    err = str(e)
                out[tag + '_dn_err'] = np.array(err)
                # sample compare: control levels over the region extended by fm on both sides
                reg_start = reg.start if reg is not None else r_data.start
                reg_size = (reg.end - reg.start) if reg is not None else n
                crng = np.random.default_rng(1000 * ci + 10 * ri + fm)
                cm = crng.normal(0, 1, reg_size + 2 * fm)
                cs = np.abs(crng.normal(0.25, 0.05, reg_size + 2 * fm)) + 0.05
                gaps = crng.random(reg_size + 2 * fm) < 0.08
                cm[gaps] = np.nan
                cs[gaps] = np.nan
                out[tag + '_sc_cm'] = cm
                out[tag + '_sc_cs'] = cs
                try:
                    pv, ps, rid = ts.compute_sample_compare_read_stats(r_data, cm, cs, fm, reg)
                    out[tag + '_sc_p'] = pv[ts.SAMP_COMP_TXT]
                    out[tag + '_sc_pos'] = ps[ts.SAMP_COMP_TXT]
                    err = ''
                except th.TomboError as e:
                    err = str(e)
                out[tag + '_sc_err'] = np.array(err)
            for std_llhr in (False, True):
                tag = 'c%d_r%d_llhr%d' % (ci, ri, int(std_llhr))
                try:
                    ll, ps, rid = ts.compute_alt_model_read_stats(r_data, std_ref, alt_refs, std_llhr, reg)
                    for name in ll:
                        out[tag + '_am_%s_v' % name] = np.asarray(ll[name], dtype=np.float64)
                        out[tag + '_am_%s_pos' % name] = np.asarray(ps[name], dtype=np.int64)
                    err = ''
                except th.TomboError as e:
                    err = str(e)
                out[tag + '_am_err'] = np.array(err)
        cases.append(dict(n=n, strand=sp['strand'], start=sp['start'], fn=fn, read_id='rid_%d' % ci,
                          regions=[None if r is None else [int(r.start), int(r.end)] for r in regs]))
    out['alt_cg'] = np.array([(k, p, m, s) for k, p, m, s in alt_ref_rows],
                             dtype=[('kmer', 'S6'), ('pos', 'i4'), ('mean', 'f8'), ('sd', 'f8')])
    out['alt_a'] = np.array([(k, p, m, s) for k, p, m, s in alt_rows_a],
                            dtype=[('kmer', 'S6'), ('pos', 'i4'), ('mean', 'f8'), ('sd', 'f8')])
    out['meta'] = np.array(json.dumps(dict(
        cases=cases, alt_models=[dict(key='alt_cg', name='5mC_CG', alt_base='C', motif='CG', mod_pos=1),
                                 dict(key='alt_a', name='6mA_GATC', alt_base='A', motif='GATC', mod_pos=2)],
        fm_offsets=[0, 1, 2])))
    path = os.path.join(HERE, 'stats_reads.npz')
    np.savez_compressed(path, **out)
    print('stats_reads.npz %.1f KB, %d arrays' % (os.path.getsize(path) / 1024., len(out)))
    errs = sorted(set(str(out[k]) for k in out if k.endswith('_err') and str(out[k])))
    print('errors seen:', errs)


def hash_code(kmer, pos):
    v = 0
    for ch in kmer:
        v = v * 4 + 'ACGT'.index(ch)
    return float(v * 7 + pos)


if __name__ == '__main__':
    main()
