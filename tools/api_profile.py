"""Where the time of resquiggle_batch(list of map_res) goes (cProfile, 2000 x 10 kb reads)."""
import os, sys, time, cProfile, pstats
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import bench
from tombo_amd import resquiggle as rq, tombo_stats as ts, tombo_helper as th

samp = th.seqSampleType('DNA', False)
model = ts.TomboModel(seq_samp_type=samp)
params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seqs, raws, dacs = bench.make_reads(np.full(n, 10000), 5, 32, 'DNA', True)
mrs = [th.resquiggleResults(align_info=th.alignInfo('r%d' % i, 'BaseCalled_template', 0, 0, 0, 0, 10000, 0),
                            genome_loc=th.genomeLocation(0, '+', 'c'), genome_seq=seqs[i], mean_q_score=10.0,
                            raw_signal=dacs[i]) for i in range(n)]
kw = dict(outlier_thresh=5.0, seq_samp_type=samp, subsample_seed=1, return_signal=os.environ.get('NO_SIGNAL') is None)
rq.resquiggle_batch(mrs[:64], model, params, **kw)
rq.resquiggle_batch(mrs, model, params, **kw)
t0 = time.perf_counter(); rq.resquiggle_batch(mrs, model, params, **kw); print('wall %.1f ms' % ((time.perf_counter() - t0) * 1e3))
if os.environ.get('API_ONE_CALL'): sys.exit(0)
pr = cProfile.Profile(); pr.enable(); rq.resquiggle_batch(mrs, model, params, **kw); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
