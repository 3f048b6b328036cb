"""Build + import the *reference* Tombo (v1.5.1) resquiggle path in THIS container only.

Test infrastructure (golden-vector generation). Nothing here ships, nothing here is
imported by the product (`tombo_amd/`), by `bench.py` or by the `-m gpu` tests:
`/root/reference` does not exist on the GPU box.

What it does:
  * cythonizes the two reference `.pyx` files *where they lie* under /root/reference
    into a scratch directory OUTSIDE the repo (default /tmp/tombo_ref_oracle) with the
    directives Cython 3 needs for this Python-2-era source (language_level=2, cpow=True);
  * makes a scratch package `tombo` whose __path__ is [scratch, /root/reference/tombo], so the
    reference's own .py files are imported in place (never copied) next to the built .so files;
  * installs four import shims (h5py stub, mappy stub, np.NAN, errstate round
    scipy.stats.halfnorm.expect) and returns the imported reference modules.

No reference source, bytecode or binary ever enters /root/repo.
"""
import os
import sys
import types
import subprocess

REF_ROOT = os.environ.get('TOMBO_REFERENCE', '/root/reference')
SCRATCH = os.environ.get('TOMBO_REF_SCRATCH', '/tmp/tombo_ref_oracle')

_SETUP = r'''
import sys, numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
ref = sys.argv.pop(1)
exts = [Extension('tombo._c_dynamic_programming', [ref + '/tombo/_c_dynamic_programming.pyx'],
                  include_dirs=[numpy.get_include()], language='c++',
                  extra_compile_args=['-O2']),
        Extension('tombo._c_helper', [ref + '/tombo/_c_helper.pyx'],
                  include_dirs=[numpy.get_include()], language='c++',
                  extra_compile_args=['-O2'])]
setup(name='tombo_ref_oracle', ext_modules=cythonize(
    exts, build_dir='cy_build',
    compiler_directives={'language_level': 2, 'cpow': True, 'embedsignature': True}))
'''


def build(force=False):
    pkg = os.path.join(SCRATCH, 'tombo')
    have = os.path.isdir(pkg) and sum(f.endswith('.so') for f in os.listdir(pkg)) >= 2
    if have and not force:
        return pkg
    os.makedirs(pkg, exist_ok=True)
    with open(os.path.join(SCRATCH, 'setup_ref.py'), 'w') as fp:
        fp.write(_SETUP)
    with open(os.path.join(pkg, '__init__.py'), 'w') as fp:
        fp.write("__path__ = [%r, %r]\n" % (pkg, os.path.join(REF_ROOT, 'tombo')))
    subprocess.check_call(
        [sys.executable, 'setup_ref.py', REF_ROOT, 'build_ext', '--inplace',
         '--build-temp', 'cy_tmp'], cwd=SCRATCH,
        stdout=subprocess.DEVNULL)
    return pkg


def _install_shims():
    import numpy as np
    if 'h5py' not in sys.modules:
        h5 = types.ModuleType('h5py')

        class File(object):
            pass
        h5.File = File
        sys.modules['h5py'] = h5
    if 'mappy' not in sys.modules:
        mp = types.ModuleType('mappy')

        class Aligner(object):
            def __init__(self, *a, **k):
                pass

            def seq(self, *a, **k):
                return None

        class ThreadBuffer(object):
            pass
        mp.Aligner = Aligner
        mp.ThreadBuffer = ThreadBuffer
        sys.modules['mappy'] = mp
    if not hasattr(np, 'NAN'):
        np.NAN = np.nan
    from scipy import stats
    if not getattr(stats.halfnorm, '_tombo_shim', False):
        orig = stats.halfnorm.expect

        def expect(*a, **k):
            with np.errstate(all='ignore'):
                return orig(*a, **k)
        stats.halfnorm.expect = expect
        stats.halfnorm._tombo_shim = True


def load():
    """Returns (resquiggle, tombo_stats, tombo_helper) reference modules."""
    build()
    _install_shims()
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from tombo import resquiggle as rq, tombo_stats as ts, tombo_helper as th
    return rq, ts, th


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    rq, ts, th = load()
    print('reference tombo imported from', rq.__file__)
    print('HALF_NORM_EXPECTED_VAL', float(ts.HALF_NORM_EXPECTED_VAL).hex())
