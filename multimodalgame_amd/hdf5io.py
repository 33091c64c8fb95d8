"""Minimal HDF5 reader/writer over ctypes -> libhdf5 (h5py is not installable in this image).

Covers exactly the feature-file schema of the reference (written by
utils/package_data.py:238-243, read by misc.py:257-302): 1-D integer `Target`, 1-D fixed-length
byte-string `Location`, float32 N-D feature datasets.  Whole datasets are read at once (the 30-class
files are a few MB); the reference's per-batch reopen + fancy-index read is a host-side inefficiency,
not a semantic (misc.py:284-300)."""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = ["/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5"), "libhdf5.so", "libhdf5_serial.so"]
_lib = None


class Hdf5Error(IOError):
    pass


def lib():
    global _lib
    if _lib is None:
        for c in _CANDIDATES:
            if not c:
                continue
            try:
                _lib = C.CDLL(c)
                break
            except OSError:
                continue
        if _lib is None:
            raise Hdf5Error("libhdf5 not found (tried %s)" % _CANDIDATES)
        L = _lib
        L.H5open()
        i64 = C.c_int64
        for name, res, args in [
            ("H5Fopen", i64, [C.c_char_p, C.c_uint, i64]), ("H5Fcreate", i64, [C.c_char_p, C.c_uint, i64, i64]),
            ("H5Fclose", C.c_int, [i64]), ("H5Dopen2", i64, [i64, C.c_char_p, i64]),
            ("H5Dcreate2", i64, [i64, C.c_char_p, i64, i64, i64, i64, i64]), ("H5Dclose", C.c_int, [i64]),
            ("H5Dget_space", i64, [i64]), ("H5Dget_type", i64, [i64]),
            ("H5Dread", C.c_int, [i64, i64, i64, i64, i64, C.c_void_p]),
            ("H5Dwrite", C.c_int, [i64, i64, i64, i64, i64, C.c_void_p]),
            ("H5Sget_simple_extent_ndims", C.c_int, [i64]),
            ("H5Sget_simple_extent_dims", C.c_int, [i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ("H5Screate_simple", i64, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ("H5Sclose", C.c_int, [i64]), ("H5Tget_class", C.c_int, [i64]), ("H5Tget_size", C.c_size_t, [i64]),
            ("H5Tcopy", i64, [i64]), ("H5Tset_size", C.c_int, [i64, C.c_size_t]), ("H5Tclose", C.c_int, [i64]),
            ("H5Lexists", C.c_int, [i64, C.c_char_p, i64]), ("H5Eset_auto2", C.c_int, [i64, C.c_void_p, C.c_void_p]),
            ("H5Tcreate", i64, [C.c_int, C.c_size_t]), ("H5Tinsert", C.c_int, [i64, C.c_char_p, C.c_size_t, i64]),
            ("H5Tarray_create2", i64, [i64, C.c_uint, C.POINTER(C.c_uint64)]),
        ]:
            f = getattr(L, name); f.restype = res; f.argtypes = args
        L.H5Eset_auto2(0, None, None)          # errors are reported through return codes -> exceptions
    return _lib


def _g(name):
    return C.c_int64.in_dll(lib(), name).value


H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND = 0, 1, 3, 6


def _compound_type(dtype):
    """HDF5 compound type mirroring a numpy structured dtype whose fields are S<n>, <i4, <f4 or <f4 sub-arrays
    (the record layouts of binary_vectors.py:24-46).  Returns (type id, ids to close afterwards)."""
    L = lib()
    tid = L.H5Tcreate(H5T_COMPOUND, dtype.itemsize)
    owned = [tid]
    for name in dtype.names:
        ft, off = dtype.fields[name][0], dtype.fields[name][1]
        base, shape = (ft.subdtype if ft.subdtype else (ft, ()))
        if base.kind == "S":
            m = L.H5Tcopy(_g("H5T_C_S1_g")); L.H5Tset_size(m, base.itemsize); owned.append(m)
        elif base.kind in "iu" and base.itemsize == 4:
            m = _g("H5T_NATIVE_INT_g")
        elif base.kind == "f" and base.itemsize == 4:
            m = _g("H5T_NATIVE_FLOAT_g")
        else:
            raise Hdf5Error("unsupported field %s: %s" % (name, ft))
        if shape:
            dims = (C.c_uint64 * len(shape))(*shape)
            m = L.H5Tarray_create2(m, len(shape), dims); owned.append(m)
        if L.H5Tinsert(tid, name.encode(), off, m) < 0:
            raise Hdf5Error("H5Tinsert failed for " + name)
    return tid, owned


class File(object):
    def __init__(self, path, mode="r"):
        L = lib()
        self.path = os.path.expanduser(path)
        if mode == "r":
            self.fid = L.H5Fopen(self.path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self.fid = L.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError(mode)
        if self.fid < 0:
            raise Hdf5Error("cannot open %s (mode %s)" % (self.path, mode))

    def close(self):
        if self.fid >= 0:
            lib().H5Fclose(self.fid)
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __contains__(self, name):
        return lib().H5Lexists(self.fid, name.encode(), H5P_DEFAULT) > 0

    def shape(self, name):
        L = lib()
        d = L.H5Dopen2(self.fid, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        L.H5Sclose(sp); L.H5Dclose(d)
        return tuple(int(dims[i]) for i in range(nd))

    def read(self, name):
        """Whole dataset -> numpy array (float32 / int64 / fixed-length bytes)."""
        L = lib()
        d = L.H5Dopen2(self.fid, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        try:
            shape = self.shape(name)
            ft = L.H5Dget_type(d)
            cls, size = L.H5Tget_class(ft), L.H5Tget_size(ft)
            if cls == H5T_FLOAT:
                out = np.empty(shape, np.float32); mt = _g("H5T_NATIVE_FLOAT_g"); own = False
            elif cls == H5T_INTEGER:
                out = np.empty(shape, np.int64); mt = _g("H5T_NATIVE_LLONG_g"); own = False
            elif cls == H5T_STRING:
                out = np.empty(shape, "S%d" % size); mt = L.H5Tcopy(_g("H5T_C_S1_g")); L.H5Tset_size(mt, size); own = True
            else:
                raise Hdf5Error("unsupported dataset class %d for %s" % (cls, name))
            rc = L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
            if own:
                L.H5Tclose(mt)
            L.H5Tclose(ft)
            if rc < 0:
                raise Hdf5Error("H5Dread failed for " + name)
            return out
        finally:
            L.H5Dclose(d)

    def write_struct(self, name, records):
        """1-D numpy structured array -> compound-type dataset."""
        L = lib()
        records = np.ascontiguousarray(records)
        tid, owned = _compound_type(records.dtype)
        dims = (C.c_uint64 * 1)(records.shape[0])
        sp = L.H5Screate_simple(1, dims, None)
        d = L.H5Dcreate2(self.fid, name.encode(), tid, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            raise Hdf5Error("cannot create dataset " + name)
        rc = L.H5Dwrite(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, records.ctypes.data_as(C.c_void_p)) if records.shape[0] else 0
        L.H5Dclose(d); L.H5Sclose(sp)
        for t in reversed(owned):
            L.H5Tclose(t)
        if rc < 0:
            raise Hdf5Error("H5Dwrite failed for " + name)

    def read_struct(self, name, dtype):
        """Compound-type dataset -> numpy structured array of the given dtype (fields matched by name)."""
        L = lib()
        dtype = np.dtype(dtype)
        n = self.shape(name)[0]
        out = np.zeros(n, dtype)
        d = L.H5Dopen2(self.fid, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(name)
        tid, owned = _compound_type(dtype)
        rc = L.H5Dread(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) if n else 0
        L.H5Dclose(d)
        for t in reversed(owned):
            L.H5Tclose(t)
        if rc < 0:
            raise Hdf5Error("H5Dread failed for " + name)
        return out

    def write(self, name, arr):
        L = lib()
        arr = np.ascontiguousarray(arr)
        if arr.dtype.kind == "f":
            arr = arr.astype(np.float32); ft = mt = _g("H5T_NATIVE_FLOAT_g"); own = False
        elif arr.dtype.kind in "iu":
            arr = arr.astype(np.int32); ft = mt = _g("H5T_NATIVE_INT_g"); own = False
        elif arr.dtype.kind == "S":
            ft = L.H5Tcopy(_g("H5T_C_S1_g")); L.H5Tset_size(ft, arr.dtype.itemsize); mt = ft; own = True
        else:
            raise Hdf5Error("unsupported dtype %s" % arr.dtype)
        dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        sp = L.H5Screate_simple(arr.ndim, dims, None)
        d = L.H5Dcreate2(self.fid, name.encode(), ft, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            raise Hdf5Error("cannot create dataset " + name)
        rc = L.H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p))
        L.H5Dclose(d); L.H5Sclose(sp)
        if own:
            L.H5Tclose(ft)
        if rc < 0:
            raise Hdf5Error("H5Dwrite failed for " + name)
