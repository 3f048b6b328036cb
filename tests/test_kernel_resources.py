"""Register / scratch budgets of the hot kernels, read from the device ISA the build would produce
(hipcc cross-compiles without a GPU).  The workgroup-per-read kernels are latency bound and their
time follows the resident workgroups per CU in steps (k_normalize 8.5 ms at
125 VGPRs, 10.8 ms at 150); a kernel that spills to scratch loses far more.  An edit that pushes
one of them across a step should fail here, not in a profile a round later."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))

# kernel name fragment -> (max VGPRs, scratch allowed)
BUDGET = {
    '_Z4k_dpILi8ELb0EE': (128, False),        # W = 500: 4 wavefronts per SIMD
    '_Z4k_dpILi5ELb0EE': (96, False),         # W = 300: 5
    '_Z11k_normalizeIdE': (128, False),       # 2 workgroups of 512 per CU
    '_Z11k_normalizeIsE': (128, False),
    '_Z7k_peaksILi2EE': (80, False),          # 3 workgroups per CU
    '_Z7k_peaksILi5EE': (128, False),
    '_Z11k_theil_sen': (80, False),           # 3 workgroups per CU (LDS allows no more)
    '_Z14k_rescale_abszILb1EE': (96, False),        # 5 wavefronts per SIMD (LDS allows 6)
    '_Z13k_event_meansIdLi448E': (72, False),     # DNA: 7 wavefronts per SIMD
    '_Z13k_event_meansIdLi1280E': (128, False),   # RNA: 4 workgroups per CU (LDS), 4 per SIMD
    '_Z9k_main_tb': (256, False),
    '_Z10k_start_tb': (512, True),           # (np_sum's recursion stack lives in scratch: 250 values per read)
    '_Z13k_final_score': (128, False),
    '_Z9k_skip_dpILb1EE': (256, False),       # DNA: windows out of LDS (40 KB: four wavefronts per CU, whatever the registers)
    '_Z9k_skip_dpILb0EE': (128, False),
}


@pytest.fixture(scope='module')
def isa_metadata():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    from tombo_amd import _native
    flags = [f for f in _native.HIPCC_FLAGS if f not in ('-fPIC', '-shared')]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'engine.s')
        subprocess.check_call([hipcc] + flags + ['-S', '--cuda-device-only', '-o', out,
                                                 os.path.join(_native.CSRC, 'tba_engine.hip')],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    meta = {}
    for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', text, re.S):
        blk = m.group(0)
        get = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk).group(1)
        meta[get('name')] = dict(vgpr=int(get('vgpr_count')), spill=int(get('vgpr_spill_count')),
                                 scratch=int(get('private_segment_fixed_size')))
    return meta


@pytest.mark.parametrize('frag', sorted(BUDGET))
def test_kernel_fits_its_occupancy_step(isa_metadata, frag):
    hits = [k for k in isa_metadata if k.startswith(frag)]
    assert len(hits) == 1, (frag, hits)
    m = isa_metadata[hits[0]]
    max_vgpr, scratch_ok = BUDGET[frag]
    assert m['vgpr'] <= max_vgpr, (hits[0], m)
    if not scratch_ok:
        assert m['spill'] == 0 and m['scratch'] == 0, (hits[0], m)


def test_no_kernel_is_allocated_224_vgprs(isa_metadata):
    """profiles/r06_traceback_rootcause.txt: the chunk-parallel traceback computed run-dependent rows on MI355X with
    the 224 VGPRs (28 granules of 8) the compiler had given it and never with 225-256, the instructions untouched.
    What the allocation does is not understood, so no kernel of the library may land on it -- a kernel that reports
    217-224 registers needs a named clobber as in k_tb_par.h (TBP_NOT_224_VGPRS) or a different register budget."""
    bad = sorted(k for k, m in isa_metadata.items() if (m['vgpr'] + 7) // 8 == 28)
    assert not bad, bad
