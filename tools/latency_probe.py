"""Latency of a lone read: `resquiggle_read` (batch of one, 10 kb DNA, W = 500) called back to back,
and the main-DP stage time inside it -- at the clocks the idle-ish GPU picks by itself and, when
`--pin` is given, again after `rocm-smi --setperflevel high` (needs root; the setting dies with
the box).  A lone wavefront is latency bound, so its row time follows the shader clock directly."""
import os
import sys
import time
import subprocess
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from tombo_amd import _native, resquiggle as rq, synth, tombo_stats as ts, tombo_helper as th  # noqa: E402


def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                             timeout=30).stdout.decode()
        return ' | '.join(l.strip() for l in out.splitlines() if 'sclk' in l or 'mclk' in l)[:300]
    except Exception as e:
        return 'rocm-smi: %r' % (e,)


def measure(tag, mrs, model, params, samp):
    eng = rq.get_engine(0)
    for mr in mrs[:3]:
        rq.resquiggle_read(mr, model, params, 5.0, seq_samp_type=samp)
    t, dp = [], []
    for mr in mrs:
        np.random.seed(1)
        t0 = time.perf_counter()
        rq.resquiggle_read(mr, model, params, 5.0, seq_samp_type=samp)
        t.append((time.perf_counter() - t0) * 1e3)
        dp.append(dict(zip(_native.STAGE_NAMES, eng.get(_native.GET_KERNEL_MS)))['main_dp'])
    print('%-22s resquiggle_read median %.2f ms  p10 %.2f  p90 %.2f   main_dp %.2f ms   [%s]' % (
        tag, np.median(t), np.percentile(t, 10), np.percentile(t, 90), np.nanmedian(dp), clocks()))


def main():
    samp = th.seqSampleType('DNA', False)
    model = ts.TomboModel(seq_samp_type=samp)
    params = ts.load_resquiggle_parameters(samp)._replace(bandwidth=500)
    mrs = [synth.synth_map_res(model, 10000, 300 + k, **synth.DNA_SYNTH) for k in range(24)]
    measure('default clocks', mrs, model, params, samp)
    if '--pin' in sys.argv:
        r = subprocess.run(['rocm-smi', '--setperflevel', 'high'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=60)
        print('setperflevel high: rc %d %s' % (r.returncode, r.stdout.decode().strip().replace('\n', ' ')[:200]))
        time.sleep(1.0)
        measure('perf level high', mrs, model, params, samp)
        subprocess.run(['rocm-smi', '--setperflevel', 'auto'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=60)


if __name__ == '__main__':
    main()
