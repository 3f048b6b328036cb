"""Extract the two canonical k-mer level tables shipped with the reference
(tombo/tombo_models/tombo.DNA.model, tombo.RNA.180mV.model; HDF5, MPL-2.0 data) into small
.npz fixtures, without h5py (absent in this image): the `model` dataset is stored as
deflate chunks of 512 records `(S{K} kmer, <f8 level_mean, <f8 level_spread)`; `central_pos`
is an int64 attribute.  Run in the build container only.
"""
import os
import re
import sys
import zlib
import struct
import numpy as np

REF = os.environ.get('TOMBO_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tombo_amd', 'tombo_models')


def extract(fn, kmer_width):
    d = open(fn, 'rb').read()
    rec = np.dtype([('kmer', 'S%d' % kmer_width), ('mean', '<f8'), ('sd', '<f8')])
    rows = []
    for m in re.finditer(re.escape(b'\x78\x5e'), d):
        try:
            out = zlib.decompress(d[m.start():])
        except zlib.error:
            o = zlib.decompressobj()
            try:
                out = o.decompress(d[m.start():])
            except zlib.error:
                continue
        if len(out) % rec.itemsize:
            continue
        rows.append(np.frombuffer(out, dtype=rec))
    tab = np.concatenate(rows)
    i = d.find(b'central_pos\x00')
    central_pos = struct.unpack('<q', d[i + 40:i + 48])[0]
    assert tab.shape[0] == 4 ** kmer_width, tab.shape
    kmers = [k.decode() for k in tab['kmer']]
    assert kmers == sorted(kmers) and set(''.join(kmers)) == set('ACGT')
    return tab, central_pos


if __name__ == '__main__':
    for name, k in (('tombo.DNA', 6), ('tombo.RNA.180mV', 5)):
        tab, cp = extract(os.path.join(REF, 'tombo', 'tombo_models', name + '.model'), k)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), kmer=tab['kmer'],
                            mean=tab['mean'], sd=tab['sd'], central_pos=np.int64(cp))
        print(name, tab.shape, 'central_pos', cp, 'sd range', tab['sd'].min(), tab['sd'].max(),
              'mean range', tab['mean'].min(), tab['mean'].max())
