"""``scores.hdf5`` without ``h5py``: a small ``ctypes`` binding of the HDF5 C library itself.

The reference writes its evaluation scores with ``h5py`` (scripts/test_model.py:245-263): one float64
array ``[mixture, metric, which]`` per ``<checkpoint>/<test set>``, two string datasets ``metrics`` and
``which`` attached to it as HDF5 dimension scales, labels on the three axes. ``h5py`` is not in this
image, but the library it wraps is (``libhdf5`` + ``libhdf5_hl`` of the conda tree); this module calls
the handful of entry points that layout needs -- files written here are read back by ``h5py`` /
``h5dump`` / ``h5ls`` like the reference's (tests/test_scripts.py checks them with ``h5dump``).

Host-side I/O, not on the compute path. ``available()`` is False when the shared libraries cannot be
loaded; ``scripts/test_model.py`` then falls back to ``scores.npz`` as before.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_herr = ctypes.c_int
_hsize = ctypes.c_uint64
_SEARCH = ('/opt/conda/lib', '/usr/lib/x86_64-linux-gnu/hdf5/serial', '/usr/lib/x86_64-linux-gnu',
           '/usr/local/lib', '/usr/lib64')
_lib = _hl = None
_tried = False

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5T_CSET_UTF8 = 1


def _find(name):
    for env in ('BRV_HDF5_DIR', 'HDF5_DIR'):
        d = os.environ.get(env)
        if d:
            for sub in ('', 'lib'):
                p = os.path.join(d, sub, f'lib{name}.so')
                if os.path.exists(p):
                    return p
    found = ctypes.util.find_library(name)
    if found:
        return found
    for d in _SEARCH:
        p = os.path.join(d, f'lib{name}.so')
        if os.path.exists(p):
            return p
    return None


def _load():
    global _lib, _hl, _tried
    if _tried:
        return _lib is not None
    _tried = True
    try:
        p, q = _find('hdf5'), _find('hdf5_hl')
        if not p or not q:
            return False
        lib = ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        hl = ctypes.CDLL(q, mode=ctypes.RTLD_GLOBAL)
        sig = {
            'H5open': (_herr, []), 'H5Eset_auto2': (_herr, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
            'H5Fcreate': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
            'H5Fopen': (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]), 'H5Fclose': (_herr, [_hid]),
            'H5Pcreate': (_hid, [_hid]), 'H5Pclose': (_herr, [_hid]),
            'H5Pset_create_intermediate_group': (_herr, [_hid, ctypes.c_uint]),
            'H5Screate_simple': (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            'H5Sclose': (_herr, [_hid]), 'H5Sget_simple_extent_ndims': (ctypes.c_int, [_hid]),
            'H5Sget_simple_extent_dims': (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            'H5Dcreate2': (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            'H5Dopen2': (_hid, [_hid, ctypes.c_char_p, _hid]), 'H5Dclose': (_herr, [_hid]),
            'H5Dget_space': (_hid, [_hid]),
            'H5Dwrite': (_herr, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            'H5Dread': (_herr, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            'H5Dvlen_reclaim': (_herr, [_hid, _hid, _hid, ctypes.c_void_p]),
            'H5Tcopy': (_hid, [_hid]), 'H5Tset_size': (_herr, [_hid, ctypes.c_size_t]),
            'H5Tset_cset': (_herr, [_hid, ctypes.c_int]), 'H5Tclose': (_herr, [_hid]),
            'H5Lexists': (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in {
                'H5DSset_scale': (_herr, [_hid, ctypes.c_char_p]),
                'H5DSis_scale': (ctypes.c_int, [_hid]),
                'H5DSattach_scale': (_herr, [_hid, _hid, ctypes.c_uint]),
                'H5DSis_attached': (ctypes.c_int, [_hid, _hid, ctypes.c_uint]),
                'H5DSset_label': (_herr, [_hid, ctypes.c_uint, ctypes.c_char_p])}.items():
            fn = getattr(hl, name)
            fn.restype, fn.argtypes = res, args
        if lib.H5open() < 0:
            return False
        lib.H5Eset_auto2(0, None, None)           # errors come back as return codes, not on stderr
        _lib, _hl = lib, hl
        return True
    except (OSError, AttributeError):
        _lib = _hl = None
        return False


def available():
    return _load()


def _global(name):
    return _hid.in_dll(_lib, name).value


def _check(status, what):
    if status < 0:
        raise OSError(f'HDF5: {what} failed')
    return status


class File:
    """``with File(path, 'w' | 'a' | 'r') as f`` -- the subset of ``h5py.File`` the score file needs."""

    def __init__(self, path, mode='r'):
        if not _load():
            raise RuntimeError('libhdf5 / libhdf5_hl could not be loaded')
        p = os.fsencode(path)
        if mode == 'w':
            self.fid = _lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == 'a':
            self.fid = _lib.H5Fopen(p, H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(path) \
                else _lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            self.fid = _lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        _check(self.fid, f'opening {path}')

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.fid is not None:
            _lib.H5Fclose(self.fid)
            self.fid = None

    def __contains__(self, key):
        path = ''
        for part in key.strip('/').split('/'):          # H5Lexists wants every intermediate link to exist
            path = f'{path}/{part}'
            if _lib.H5Lexists(self.fid, path.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    # -- writing ---------------------------------------------------------------------------------
    def _create(self, key, type_id, shape):
        dims = (_hsize*len(shape))(*shape)
        space = _check(_lib.H5Screate_simple(len(shape), dims, None), 'H5Screate_simple')
        lcpl = _check(_lib.H5Pcreate(_global('H5P_CLS_LINK_CREATE_ID_g')), 'H5Pcreate')
        _lib.H5Pset_create_intermediate_group(lcpl, 1)
        dset = _lib.H5Dcreate2(self.fid, key.encode(), type_id, space, lcpl, H5P_DEFAULT, H5P_DEFAULT)
        _lib.H5Pclose(lcpl)
        _lib.H5Sclose(space)
        return _check(dset, f'creating {key}')

    def write_array(self, key, values):
        """float64 array; an existing dataset of the same shape is overwritten in place
        (``h5dset[...] = dset_scores``, scripts/test_model.py:249-250)."""
        a = np.ascontiguousarray(values, dtype=np.float64)
        f64 = _global('H5T_NATIVE_DOUBLE_g')
        if key in self:
            if self.read_array(key).shape != a.shape:
                raise ValueError(f'{key}: stored shape differs from {a.shape}')
            dset = _check(_lib.H5Dopen2(self.fid, key.encode(), H5P_DEFAULT), f'opening {key}')
        else:
            dset = self._create(key, f64, a.shape)
        status = _lib.H5Dwrite(dset, f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
        _lib.H5Dclose(dset)
        _check(status, f'writing {key}')

    @staticmethod
    def _vlen_str():
        t = _check(_lib.H5Tcopy(_global('H5T_C_S1_g')), 'H5Tcopy')
        _lib.H5Tset_size(t, H5T_VARIABLE)
        _lib.H5Tset_cset(t, H5T_CSET_UTF8)
        return t

    def write_strings(self, key, strings):
        """1-D dataset of variable-length UTF-8 strings (what ``h5file[key] = [str, ...]`` stores)."""
        t = self._vlen_str()
        raw = [s.encode() for s in strings]
        buf = (ctypes.c_char_p*len(raw))(*raw)
        dset = self._create(key, t, (len(raw),))
        status = _lib.H5Dwrite(dset, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, ctypes.cast(buf, ctypes.c_void_p))
        _lib.H5Dclose(dset)
        _lib.H5Tclose(t)
        _check(status, f'writing {key}')

    def set_dims(self, key, labels, scales):
        """Axis labels and dimension scales of dataset ``key``: ``scales = {axis: scale dataset key}``
        (``h5dset.dims[i].label = ...``, ``h5dset.dims[i].attach_scale(...)``)."""
        dset = _check(_lib.H5Dopen2(self.fid, key.encode(), H5P_DEFAULT), f'opening {key}')
        try:
            for axis, label in enumerate(labels):
                _check(_hl.H5DSset_label(dset, axis, label.encode()), 'H5DSset_label')
            for axis, skey in scales.items():
                sc = _check(_lib.H5Dopen2(self.fid, skey.encode(), H5P_DEFAULT), f'opening {skey}')
                try:
                    if _hl.H5DSis_scale(sc) <= 0:
                        _check(_hl.H5DSset_scale(sc, None), 'H5DSset_scale')
                    if _hl.H5DSis_attached(dset, sc, axis) <= 0:
                        _check(_hl.H5DSattach_scale(dset, sc, axis), 'H5DSattach_scale')
                finally:
                    _lib.H5Dclose(sc)
        finally:
            _lib.H5Dclose(dset)

    # -- reading ---------------------------------------------------------------------------------
    def _shape(self, dset):
        space = _lib.H5Dget_space(dset)
        n = _lib.H5Sget_simple_extent_ndims(space)
        dims = (_hsize*max(n, 1))()
        _lib.H5Sget_simple_extent_dims(space, dims, None)
        _lib.H5Sclose(space)
        return tuple(int(d) for d in dims[:n])

    def read_array(self, key):
        dset = _check(_lib.H5Dopen2(self.fid, key.encode(), H5P_DEFAULT), f'opening {key}')
        out = np.empty(self._shape(dset), dtype=np.float64)
        status = _lib.H5Dread(dset, _global('H5T_NATIVE_DOUBLE_g'), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                              out.ctypes.data_as(ctypes.c_void_p))
        _lib.H5Dclose(dset)
        _check(status, f'reading {key}')
        return out

    def read_strings(self, key):
        dset = _check(_lib.H5Dopen2(self.fid, key.encode(), H5P_DEFAULT), f'opening {key}')
        (n,) = self._shape(dset)
        t = self._vlen_str()
        buf = (ctypes.c_char_p*n)()
        status = _lib.H5Dread(dset, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, ctypes.cast(buf, ctypes.c_void_p))
        out = [b.decode() if b is not None else '' for b in buf] if status >= 0 else None
        if status >= 0:
            space = _lib.H5Dget_space(dset)
            _lib.H5Dvlen_reclaim(t, space, H5P_DEFAULT, ctypes.cast(buf, ctypes.c_void_p))
            _lib.H5Sclose(space)
        _lib.H5Tclose(t)
        _lib.H5Dclose(dset)
        _check(status, f'reading {key}')
        return out
