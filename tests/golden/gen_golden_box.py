"""Golden vectors over the PARAMETER BOX the GPU fuzz explores (tests/test_gpu_param_fuzz.py:
--segmentation-parameters / --signal-align-parameters overrides, outlier_thresh, max_raw_cpts,
skip_seq_scaling, const_scale; reads that disagree with their sequence), from the live reference
(build container only):

    python tests/golden/gen_golden_box.py        # writes tests/golden/o_box_*.npz

These fixtures pin the ORACLE only (tests/test_oracle_golden.py): the engine is compared with the
oracle over the same box by hypothesis on the GPU, so reference -> oracle -> engine is closed over
the box without a GPU test per fixture.  Float arrays are kept as SHA-256 + length (`lite`).
Points at which the reference dies with a non-Tombo exception are skipped.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

N_PARAM, N_EDIT, SEED = 28, 20, 20260928


def pick(rng, xs):
    return xs[int(rng.integers(0, len(xs)))]


def param_point(rng):
    rna = bool(rng.integers(0, 2))
    seg = (int(rng.integers(2, 41)), int(rng.integers(1, 10)), int(rng.integers(1, 4)), int(rng.integers(3, 21)))
    bw = pick(rng, [60, 100, 128, 200, 256, 300, 320, 321, 500, 700, 1000])
    start_bw = pick(rng, [200, 400, 750, 1000])
    aln = (float(rng.uniform(2.0, 7.0)), float(rng.uniform(1.0, 7.0)), bw, 1500,
           pick(rng, [2.0, 5.0, 10.0, 20.0, 50.0]), int(rng.integers(0, 61)), start_bw,
           pick(rng, [start_bw, 1500, 2500]), pick(rng, [60, 150, 250]))
    outlier = pick(rng, [None, 2.0, 3.0, 5.0, 8.0])
    const_scale = pick(rng, [None, None, 9.0])
    if rna and outlier is None and const_scale is None:
        outlier = 4.0     # a TypeError in the reference (tombo_stats.py:228)
    return dict(samp_name='RNA' if rna else 'DNA', seg_params=seg, sig_aln_params=aln, outlier_thresh=outlier,
                max_raw_cpts=pick(rng, [None, 3, 30, 200]), skip_seq_scaling=bool(rng.integers(0, 2)),
                const_scale=const_scale, seed=int(rng.integers(0, 10 ** 6)),
                n_bases=pick(rng, [120, 300, 700, 1300]))


def edit_point(rng):
    from tombo_amd import synth
    rna = bool(rng.integers(0, 2))
    kind = pick(rng, ['cut', 'insert', 'truncate', 'none'])
    edit = None
    if kind == 'cut':
        edit = dict(kind='cut', n=int(rng.integers(3, 91)))
    elif kind == 'insert':
        edit = dict(kind='insert', n=int(rng.integers(3, 91)), seed=int(rng.integers(0, 100)))
    elif kind == 'truncate':
        edit = dict(kind='truncate', frac=pick(rng, [0.5, 0.8, 0.9, 0.95, 0.99]))
    kw = dict(synth.RNA_SYNTH if rna else synth.DNA_SYNTH, noise_sd=pick(rng, [0.15, 0.25, 0.5, 0.9]))
    dwell = pick(rng, [None, 0.25, 0.4, 1.6, 3.0])
    if dwell is not None:
        kw['mean_dwell'] = max(2, int(kw['mean_dwell'] * dwell))
        kw['min_dwell'] = max(1, min(kw['min_dwell'], kw['mean_dwell'] // 2))
    lead = pick(rng, [None, 20, 2500, 7000])
    if lead is not None:
        kw['lead'] = lead
    return dict(samp_name='RNA' if rna else 'DNA', edit=edit, synth_kw=kw, seed=int(rng.integers(0, 10 ** 6)),
                n_bases=pick(rng, [260, 600, 1100, 1700]), bandwidth=pick(rng, [100, 300, 500]),
                band_bound_thresh=pick(rng, [10, 40]))


def main():
    rng = np.random.default_rng(SEED)
    pts = [param_point(rng) for _ in range(N_PARAM)] + [edit_point(rng) for _ in range(N_EDIT)]
    n_ok = n_skip = 0
    for k, pt in enumerate(pts):
        name = 'o_box_%02d' % k
        try:
            gg.run_case(name=name, lite=True, **pt)
            n_ok += 1
        except gg.th.TomboError:
            raise
        except Exception as e:      # the reference's "unexpected error": nothing to pin
            n_skip += 1
            path = os.path.join(HERE, name + '.npz')
            if os.path.exists(path):
                os.remove(path)
            print('%-12s skipped: %s: %s' % (name, type(e).__name__, str(e)[:80]))
    print('%d fixtures, %d points skipped' % (n_ok, n_skip))


if __name__ == '__main__':
    main()
