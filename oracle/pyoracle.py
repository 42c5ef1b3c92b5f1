"""ctypes binding of the CPU oracle (oracle/vsom_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path never imports this module.
PARITY UNPINNED (see vsom_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvsom_oracle.so")

STANDARD, MEDIAN, CLR = 0, 1, 2
EXPONENTIAL, INVERSE_PROPORTIONAL, BATCHMAP = 0, 1, 2


def build(force=False):
    """Compile libvsom_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "vsom_oracle.c")
    hdr = os.path.join(_HERE, "vsom_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libvsom_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Som(C.Structure):
    _fields_ = [("width", C.c_size_t), ("height", C.c_size_t), ("depth", C.c_size_t),
                ("in_len", C.c_size_t), ("transform", C.c_int),
                ("map", C.POINTER(C.c_float)), ("sigma", C.POINTER(C.c_float)),
                ("S", C.POINTER(C.c_float)), ("weight", C.POINTER(C.c_float)),
                ("hits", C.POINTER(C.c_uint64))]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp, u64p, szp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)
    sp = C.POINTER(_Som)
    L.vso_length.restype = C.c_size_t
    L.vso_length.argtypes = [C.c_int, C.c_size_t]
    L.vso_create.restype = sp
    L.vso_create.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    L.vso_free.argtypes = [sp]
    L.vso_random_initialize.argtypes = [sp, C.c_int, C.c_float]
    L.vso_comparer_len.restype = C.c_size_t
    L.vso_comparer_len.argtypes = [C.c_int, C.c_size_t]
    L.vso_comparer.argtypes = [C.c_int, fp, C.c_size_t, fp, C.c_size_t, fp]
    L.vso_stepper.argtypes = [C.c_int, fp, C.c_size_t, fp, C.c_size_t, fp]
    L.vso_dot_self.restype = C.c_float
    L.vso_dot_self.argtypes = [fp, C.c_size_t]
    L.vso_somindex.argtypes = [sp, C.c_size_t, szp, szp]
    L.vso_dist.restype = C.c_double
    L.vso_dist.argtypes = [sp, C.c_size_t, fp]
    L.vso_find_bmu.restype = C.c_size_t
    L.vso_find_bmu.argtypes = [sp, fp]
    L.vso_find_local_bmu.restype = C.c_size_t
    L.vso_find_local_bmu.argtypes = [sp, fp, C.c_size_t]
    L.vso_neighbourhood_weight.restype = C.c_double
    L.vso_neighbourhood_weight.argtypes = [C.c_size_t] * 4 + [C.c_double]
    L.vso_batch_phase1_range.argtypes = [sp, fp, C.c_size_t, C.c_size_t, C.c_size_t, u64p, fp,
                                         C.c_int, C.c_int]
    L.vso_batch_phase1_finish.restype = C.c_float
    L.vso_batch_phase1_finish.argtypes = [sp, u64p, fp, C.c_size_t]
    L.vso_batch_phase2_range.argtypes = [sp, fp, C.c_size_t, u64p, C.c_double, C.c_size_t,
                                         C.c_size_t, C.c_int]
    L.vso_batch_epoch.restype = C.c_float
    L.vso_batch_epoch.argtypes = [sp, fp, C.c_size_t, u64p, C.c_double, C.c_int, C.c_int]
    L.vso_batch_epoch_faithful.restype = C.c_float
    L.vso_batch_epoch_faithful.argtypes = [sp, fp, C.c_size_t, u64p, C.c_double, C.c_int]
    L.vso_train_batch.restype = C.c_size_t
    L.vso_train_batch.argtypes = [sp, fp, szp, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                  fp, C.c_int]
    L.vso_train_single.restype = C.c_size_t
    L.vso_train_single.argtypes = [sp, fp, C.c_double, C.c_double, u64p, C.c_int, fp, fp]
    L.vso_train_online_chunk.restype = C.c_float
    L.vso_train_online_chunk.argtypes = [sp, fp, C.c_size_t, u64p, C.c_double, C.c_double, C.c_int]
    L.vso_train_online_chunk_from.restype = C.c_float
    L.vso_train_online_chunk_from.argtypes = [sp, fp, C.c_size_t, u64p, C.c_double, C.c_double, C.c_int, C.c_float]
    L.vso_train_online.argtypes = [sp, fp, szp, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_int, fp]
    L.vso_find_restricted_bmu.restype = C.c_size_t
    L.vso_find_restricted_bmu.argtypes = [sp, fp, C.c_uint64]
    L.vso_find_restricted_bmd.argtypes = [sp, fp, C.c_uint64, C.POINTER(C.c_double)]
    L.vso_dist_raw.restype = C.c_double
    L.vso_dist_raw.argtypes = [sp, C.c_size_t, fp]
    L.vso_update_umatrix.argtypes = [sp, C.POINTER(C.c_double)]
    L.vso_max_threads.restype = C.c_int
    _lib = L
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def max_threads():
    return int(lib().vso_max_threads())


def length(transform, in_len):
    return int(lib().vso_length(transform, in_len))


def comparer(transform, value, model):
    value, model = _as_f32(value), _as_f32(model)
    n = int(lib().vso_comparer_len(transform, model.size))
    out = np.empty(n, np.float32)
    lib().vso_comparer(transform, _f(value), value.size, _f(model), model.size, _f(out))
    return out


def stepper(transform, value, model):
    value, model = _as_f32(value), _as_f32(model)
    out = np.empty(model.size, np.float32)
    lib().vso_stepper(transform, _f(value), value.size, _f(model), model.size, _f(out))
    return out


def dot_self(r):
    r = _as_f32(r)
    return np.float32(lib().vso_dot_self(_f(r), r.size))


def neighbourhood_weight(cx, cy, bx, by, sigma):
    return float(lib().vso_neighbourhood_weight(cx, cy, bx, by, float(sigma)))


class OracleSom:
    """State of `class Som` held by the C oracle; numpy views alias the C arrays."""

    def __init__(self, width, height, in_len, transform=STANDARD):
        self._p = lib().vso_create(width, height, in_len, transform)
        if not self._p:
            raise MemoryError("vso_create failed")
        s = self._p.contents
        self.width, self.height = int(s.width), int(s.height)
        self.depth, self.in_len, self.transform = int(s.depth), int(s.in_len), int(s.transform)
        n, d = self.width * self.height, self.depth
        self.n_nodes = n
        self.map = np.ctypeslib.as_array(s.map, shape=(n, d))
        self.sigma = np.ctypeslib.as_array(s.sigma, shape=(n, d))
        self.S = np.ctypeslib.as_array(s.S, shape=(n, d))
        self.weight = np.ctypeslib.as_array(s.weight, shape=(n,))
        self.hits = np.ctypeslib.as_array(s.hits, shape=(n,))

    def close(self):
        if self._p:
            self.map = self.sigma = self.S = self.weight = self.hits = None
            lib().vso_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state helpers -------------------------------------------------
    def set_state(self, map=None, sigma=None, S=None, weight=None, hits=None):
        if map is not None:
            self.map[...] = np.asarray(map, np.float32).reshape(self.map.shape)
        if sigma is not None:
            self.sigma[...] = np.asarray(sigma, np.float32).reshape(self.sigma.shape)
        if S is not None:
            self.S[...] = np.asarray(S, np.float32).reshape(self.S.shape)
        if weight is not None:
            self.weight[...] = np.asarray(weight, np.float32).reshape(self.weight.shape)
        if hits is not None:
            self.hits[...] = np.asarray(hits, np.uint64).reshape(self.hits.shape)

    def random_initialize(self, seed, sigma):
        lib().vso_random_initialize(self._p, int(seed), float(sigma))

    def somindex(self, idx):
        x, y = C.c_size_t(), C.c_size_t()
        lib().vso_somindex(self._p, idx, C.byref(x), C.byref(y))
        return int(x.value), int(y.value)

    # --- search ---------------------------------------------------------
    def dist(self, node, v):
        v = _as_f32(v)
        return float(lib().vso_dist(self._p, node, _f(v)))

    def find_bmu(self, v):
        v = _as_f32(v)
        return int(lib().vso_find_bmu(self._p, _f(v)))

    def find_local_bmu(self, v, last):
        v = _as_f32(v)
        return int(lib().vso_find_local_bmu(self._p, _f(v), int(last)))

    # --- consumers of the search / U-matrix -------------------
    def find_restricted_bmu(self, v, min_hits):
        v = _as_f32(v)
        return int(lib().vso_find_restricted_bmu(self._p, _f(v), int(min_hits)))

    def find_restricted_bmd(self, v, min_hits):
        v = _as_f32(v)
        out = np.empty(self.n_nodes, np.float64)
        lib().vso_find_restricted_bmd(self._p, _f(v), int(min_hits), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def dist_raw(self, pos, v):
        v = _as_f32(v)
        return float(lib().vso_dist_raw(self._p, int(pos), _f(v)))

    def update_umatrix(self):
        out = np.empty(self.n_nodes, np.float64)
        lib().vso_update_umatrix(self._p, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    # --- batch path -----------------------------------------------------
    def _chk(self, X):
        X = _as_f32(X)
        assert X.ndim == 2 and X.shape[1] == self.in_len, (X.shape, self.in_len)
        return X

    def batch_phase1_range(self, X, s0, s1, lastbmu, sqres, is_first, nthreads=1):
        X = self._chk(X)
        lib().vso_batch_phase1_range(self._p, _f(X), X.shape[0], s0, s1, _u(lastbmu), _f(sqres),
                                     int(bool(is_first)), nthreads)

    def batch_phase1_finish(self, lastbmu, sqres):
        return np.float32(lib().vso_batch_phase1_finish(self._p, _u(lastbmu), _f(sqres),
                                                        lastbmu.size))

    def batch_phase2_range(self, X, lastbmu, sigma, n0, n1, nthreads=1):
        X = self._chk(X)
        lib().vso_batch_phase2_range(self._p, _f(X), X.shape[0], _u(lastbmu), float(sigma),
                                     n0, n1, nthreads)

    def batch_epoch(self, X, lastbmu, sigma, is_first, nthreads=1, faithful=False):
        X = self._chk(X)
        assert lastbmu.dtype == np.uint64 and lastbmu.size == X.shape[0]
        if faithful:
            return np.float32(lib().vso_batch_epoch_faithful(
                self._p, _f(X), X.shape[0], _u(lastbmu), float(sigma), int(bool(is_first))))
        return np.float32(lib().vso_batch_epoch(self._p, _f(X), X.shape[0], _u(lastbmu),
                                                float(sigma), int(bool(is_first)), nthreads))

    def train_batch(self, X, chunk_off, epochs, sigma0, sigma_decay, nthreads=1):
        X = self._chk(X)
        off = np.ascontiguousarray(chunk_off, dtype=np.uintp)
        mse = np.full(epochs, np.nan, np.float32)
        done = lib().vso_train_batch(self._p, _f(X), off.ctypes.data_as(C.POINTER(C.c_size_t)),
                                     off.size - 1, epochs, float(sigma0), float(sigma_decay),
                                     _f(mse), nthreads)
        return int(done), mse

    # --- online path ----------------------------------------------------
    def train_single(self, v, eta, sigma, last_bmu, decay_fn):
        v = _as_f32(v)
        L = int(lib().vso_comparer_len(self.transform, self.depth))
        res = np.empty(L, np.float32)
        lb = C.c_uint64(int(last_bmu))
        dist = C.c_float()
        bmu = lib().vso_train_single(self._p, _f(v), float(eta), float(sigma), C.byref(lb),
                                     int(decay_fn), _f(res), C.byref(dist))
        return int(bmu), res, np.float32(dist.value), int(lb.value)

    def train_online_chunk(self, X, lastbmu, eta, sigma, decay_fn, mse_start=0.0):
        """One chunk of trainBasicSom's sample loop; returns the epoch's running MSE accumulator after
        it (mse_start = its value before: the reference keeps one accumulator per epoch)."""
        X = self._chk(X)
        return np.float32(lib().vso_train_online_chunk_from(self._p, _f(X), X.shape[0], _u(lastbmu),
                                                            float(eta), float(sigma), int(decay_fn),
                                                            float(np.float32(mse_start))))

    def train_online(self, X, chunk_off, epochs, eta0, eta_decay, sigma0, sigma_decay, decay_fn):
        X = self._chk(X)
        off = np.ascontiguousarray(chunk_off, dtype=np.uintp)
        mse = np.full(epochs, np.nan, np.float32)
        lib().vso_train_online(self._p, _f(X), off.ctypes.data_as(C.POINTER(C.c_size_t)),
                               off.size - 1, epochs, float(eta0), float(eta_decay), float(sigma0),
                               float(sigma_decay), int(decay_fn), _f(mse))
        return mse
