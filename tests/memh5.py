"""A dict-backed stand-in for the part of the h5py group interface that
write_new_fast5_group and the FAST5 readers of the mapping step use (`__getitem__`, `values`,
`create_group`, `create_dataset`, dataset `[()]` / `[:]`, `.attrs`): the build
image has no HDF5 library, so the FAST5 writer is exercised against this and its tree is compared
with the tree the REFERENCE's writer produces on the same stand-in (gen_golden_fast5.py)."""
import numpy as np


class MemDataset(object):
    def __init__(self, data, **kw):
        self.data = np.array(data)
        self.kw = kw
        self.attrs = {}

    def __getitem__(self, key):          # ds[()] -> scalar / whole array, ds[:] -> array
        v = self.data[key]
        return v.item() if isinstance(v, np.ndarray) and v.shape == () and v.dtype.kind in 'SUO' else v


class MemGroup(object):
    def __init__(self):
        self.items = {}
        self.attrs = {}

    def create_group(self, name):
        g = self
        parts = [p for p in name.split('/') if p]
        for i, part in enumerate(parts):
            if part not in g.items:
                g.items[part] = MemGroup()
            elif i == len(parts) - 1:
                raise ValueError('Unable to create group (name already exists)')   # as h5py
            g = g.items[part]
        return g

    def __delitem__(self, name):
        del self.items[name]

    def create_dataset(self, name, data=None, **kw):
        d = MemDataset(data, **kw)
        self.items[name] = d
        return d

    def __getitem__(self, path):
        g = self
        for part in [p for p in path.split('/') if p]:
            g = g.items[part]
        return g

    def values(self):
        return self.items.values()

    def __contains__(self, name):
        return name in self.items


def tree(node, prefix=''):
    """flat {path: value} view of attributes and datasets"""
    out = {}
    for k, v in node.attrs.items():
        out[prefix + '@' + k] = v
    if isinstance(node, MemGroup):
        for name, child in node.items.items():
            if isinstance(child, MemDataset):
                out[prefix + '/' + name] = child.data
                for k, v in child.attrs.items():
                    out[prefix + '/' + name + '@' + k] = v
            else:
                out.update(tree(child, prefix + '/' + name))
    return out
