"""A ~100-line stand-in for the parts of xarray.DataArray that xhistogram's wrapper touches.

TEST DOUBLE ONLY: xarray is not installed in the build image.  It models named
dims, dimension/non-dimension coordinates, attrs, ``reset_coords(drop=True)``, ``expand_dims``,
``transpose``, ``get_axis_num``, ``__getitem__`` of a coordinate and ``xr.align(join="exact")``
with xarray's documented semantics, nothing more.  When the real package is importable the tests
use it instead (tests/test_xarray_wrapper.py).
"""
import numpy as np

__version__ = "0.0-double"


class _Coords(dict):
    pass


class DataArray:
    def __init__(self, data, dims=None, coords=None, name=None, attrs=None):
        self.data = data
        self.dims = tuple(dims) if dims is not None else tuple("dim_%d" % i for i in range(np.ndim(data)))
        assert len(self.dims) == np.ndim(data), (self.dims, np.shape(data))
        self.name = name
        self.attrs = dict(attrs or {})
        self.coords = _Coords()
        for k, v in (coords or {}).items():
            if isinstance(v, DataArray):
                self.coords[k] = DataArray(v.data, v.dims, None, k, v.attrs)
            elif isinstance(v, tuple):
                cd, cv = v[0], np.asarray(v[1])
                self.coords[k] = DataArray(cv, cd, None, k, v[2] if len(v) > 2 else None)
            else:
                self.coords[k] = DataArray(np.asarray(v), (k,), None, k)
        for k, c in self.coords.items():
            for d, n in zip(c.dims, np.shape(c.data)):
                assert d in self.dims and self.sizes[d] == n, "coordinate %r does not fit" % k

    @property
    def shape(self):
        return tuple(np.shape(self.data))

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def values(self):
        d = self.data
        return d.cpu().numpy() if hasattr(d, "cpu") else np.asarray(d)

    def __getitem__(self, key):
        return self.coords[key]

    def get_axis_num(self, dim):
        return self.dims.index(dim)

    def reset_coords(self, drop=False):
        assert drop
        keep = {k: v for k, v in self.coords.items() if k in self.dims}
        return DataArray(self.data, self.dims, keep, self.name, self.attrs)

    def expand_dims(self, mapping):
        data = self.data
        for _ in mapping:
            data = data[None]
        return DataArray(data, tuple(mapping) + self.dims, self.coords, self.name, self.attrs)

    def transpose(self, *dims):
        perm = [self.dims.index(d) for d in dims]
        data = self.data.permute(*perm) if hasattr(self.data, "permute") else np.transpose(self.data, perm)
        return DataArray(data, dims, self.coords, self.name, self.attrs)


def align(*objs, join="exact"):
    assert join == "exact"
    for a in objs:
        for b in objs:
            for d in set(a.dims) & set(b.dims):
                if a.sizes[d] != b.sizes[d]:
                    raise ValueError("cannot align objects with join='exact': dimension %r differs" % d)
                if d in a.coords and d in b.coords and not np.array_equal(a.coords[d].values, b.coords[d].values):
                    raise ValueError("cannot align objects with join='exact': index %r differs" % d)
    return objs
