"""TEST INFRASTRUCTURE: a stand-in for tombo_amd._native.Engine / PinnedArray so that the HOST
logic of bench.py and the streaming pipeline (work queue, slots, feeder, per-rank report, JSON
line) can run on a box without a GPU.  It computes nothing: every read "succeeds" with made-up
boundaries.  bench.py loads it through its hidden --engine-stub option (tests only)."""
import time

import numpy as np


class StubPinned(object):
    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype)
        self.nbytes = self.a.nbytes

    def close(self):
        self.a = None


def install(_native):
    K = 6

    class StubEngine(object):
        def __init__(self, device=0):
            self.device = device
            self.kmer_width = K
            self._stage = None

        def set_model(self, *a):
            pass

        def ensure_model(self, m):
            pass

        def device_mem(self):
            return 64 << 30, 64 << 30

        def held_bytes(self):
            return 0

        def set_sharing(self, n_engines):
            pass

        def host_stage(self):
            if self._stage is None:
                self._stage = _native.PinnedStage()
            return self._stage

        def upload(self, params, opts, raws, seqs, **kw):
            raw_off = np.zeros(len(raws) + 1, np.int64)
            np.cumsum([len(r) for r in raws], out=raw_off[1:])
            seq_off = np.zeros(len(seqs) + 1, np.int64)
            np.cumsum([len(s) for s in seqs], out=seq_off[1:])
            self.upload_packed(params, opts, None, raw_off, None, seq_off)

        def upload_packed(self, params, opts, raw, raw_off, seq, seq_off, **kw):
            self.n = len(raw_off) - 1
            self.raw_off, self.seq_off = np.asarray(raw_off), np.asarray(seq_off)
            self.B = np.maximum(np.diff(self.seq_off) - K + 1, 0)
            self.ref_off = np.concatenate([[0], np.cumsum(self.B)]).astype(np.int64)
            self.seg_off = self.ref_off + np.arange(self.n + 1)
            self.n_raw_total = int(self.raw_off[-1])

        def stats(self):
            return 3.0e6 * self.n, 5.0e6 * self.n

        def enqueue(self):
            time.sleep(0.002)

        run = enqueue

        def sync(self):
            pass

        def get(self, what):
            ms = np.zeros(32, np.float32)
            ms[_native.STAGE_NAMES.index('main_dp')] = 1.0
            ms[15] = 2.0
            return ms

        def download(self, want_norm=True):
            return dict(status=np.zeros(self.n, np.int32))

        def download_async(self, results=None, segs32=None, segs64=None, norm=None):
            if results is not None:
                results['status'][:self.n] = 0

        def close(self):
            pass

    class StubSynth(object):
        """_native.Synth without a device: offsets only (lead 200 + 9 samples per base + trail 100)"""
        calls = []

        def __init__(self, std_ref, device=0):
            self.device = device

        def generate(self, sp, seed, n_bases, raw_dtype=np.int16, first_read=0):
            nb = np.asarray(n_bases, np.int64)
            StubSynth.calls.append((int(seed), int(first_read), int(nb.shape[0])))
            raw_off = np.concatenate([[0], np.cumsum(300 + 9 * nb)]).astype(np.int64)
            seq_off = np.concatenate([[0], np.cumsum(nb + K - 1)]).astype(np.int64)
            return None, raw_off, None, seq_off

        def close(self):
            pass

    _native.Engine = StubEngine
    _native.PinnedArray = StubPinned
    _native.Synth = StubSynth
