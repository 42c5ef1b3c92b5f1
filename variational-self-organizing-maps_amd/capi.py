"""ctypes binding of libvsom_hip.so (include/vsom_hip.h).

This is the product path: it loads the in-tree HIP library and fails loudly when the
library or a gfx950 device is missing -- there is no CPU fallback and nothing here touches
oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree library; VSOM_LIB names another BUILD OF THE SAME LIBRARY (the -DVSOM_DEVELOPMENT build of tools/exp/,
# which times variant code objects) without overwriting the shipped file
_INTREE = os.path.join(_HERE, "libvsom_hip.so")
LIB_PATH = os.environ.get("VSOM_LIB") or _INTREE

STANDARD, MEDIAN, CLR = 0, 1, 2
EXPONENTIAL, INVERSE_PROPORTIONAL, BATCHMAP = 0, 1, 2
BMU_AUTO, BMU_EXACT, BMU_SHORTLIST = 0, 1, 2
UPDATE_STRICT, UPDATE_FMA, UPDATE_FMA_SIGMA = 0, 1, 2
BUF_MAP, BUF_SIGMA, BUF_S, BUF_WEIGHT, BUF_HITS, BUF_LASTBMU, BUF_SQRES, BUF_CHUNK = range(8)
T_STAGE, T_BMU, T_FINISH, T_CW, T_UPDATE, T_ONLINE, T_SIGMA, T_COUNT = range(8)
TIMER_NAMES = ["stage", "bmu", "finish", "cw", "update", "online", "sigma"]

# every symbol include/vsom_hip.h declares (tests/test_capi_symbols.py checks the header too)
SYMBOLS = [
    "vsom_last_error", "vsom_device_count", "vsom_create", "vsom_destroy", "vsom_set_stream",
    "vsom_synchronize", "vsom_set_bmu_mode", "vsom_set_update_mode", "vsom_set_column_compaction", "vsom_set_row_dedupe", "vsom_get_shortlist_stats", "vsom_depth", "vsom_nodes", "vsom_set_state",
    "vsom_get_state", "vsom_upload_chunk", "vsom_set_chunk_device", "vsom_host_alloc", "vsom_host_free",
    "vsom_prefetch_chunk", "vsom_prefetch_wait", "vsom_commit_chunk", "vsom_stage_next_device", "vsom_get_last_bmu",
    "vsom_set_last_bmu", "vsom_get_sqres", "vsom_bmu_batch", "vsom_find_bmu", "vsom_dist_single", "vsom_find_local_bmu", "vsom_find_restricted_bmu", "vsom_distances_single", "vsom_bmu_local_batch",
    "vsom_distances", "vsom_bmu_restricted_batch", "vsom_distances_row", "vsom_distances_raw", "vsom_batch_phase1_async", "vsom_batch_finish_async",
    "vsom_batch_phase2_async", "vsom_batch_epoch_async", "vsom_batch_epoch", "vsom_get_mse",
    "vsom_residual_len", "vsom_train_single", "vsom_train_online_chunk", "vsom_train_online_chunk_acc", "vsom_train_online_chunk_fetch", "vsom_upload_chunk_async", "vsom_get_online_search_stats",
    "vsom_neighbourhood_weight", "vsom_device_ptr", "vsom_chunk_size", "vsom_pitch",
    "vsom_chunk_pitch", "vsom_small_map_chains", "vsom_enable_timing", "vsom_enable_timing_of", "vsom_get_timing",
    "vsom_group_create", "vsom_group_destroy", "vsom_group_size", "vsom_group_ctx", "vsom_group_transport",
    "vsom_group_synchronize", "vsom_group_set_state", "vsom_group_get_state", "vsom_group_set_update_mode",
    "vsom_group_set_bmu_mode", "vsom_group_upload_chunk", "vsom_group_prefetch_chunk", "vsom_group_prefetch_wait",
    "vsom_group_commit_chunk", "vsom_group_set_chunk_device", "vsom_group_set_last_bmu", "vsom_group_get_last_bmu",
    "vsom_group_batch_epoch_async", "vsom_group_batch_epoch", "vsom_group_get_mse",
]


class VsomError(RuntimeError):
    pass


def has_contracted(transform):
    """whether vsom_set_update_mode(VSOM_UPDATE_FMA) changes the chain arithmetic of this transformation
    (include/vsom_hip.h, vsom_update_mode)"""
    return int(transform) == STANDARD      # Median (exact fused operations) and CLR have one arithmetic


def build(force=False):
    """Compile libvsom_hip.so for gfx950 with csrc/build.sh (hipcc cross-compiles on CPU)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(_HERE, "csrc")):
            if f.endswith(".o"):
                os.remove(os.path.join(_HERE, "csrc", f))
    subprocess.check_call(["bash", script, _INTREE], stdout=subprocess.DEVNULL)
    return _INTREE


def hip_runtimes():
    """paths of the libamdhip64 copies mapped into this process (/proc/self/maps)"""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                at = line.find("/")
                if at >= 0 and "libamdhip64.so" in line:
                    found.add(line[at:].strip())
    except OSError:
        pass
    return sorted(found)


def assert_single_hip_runtime(what):
    """PyTorch's wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Loaded FIRST, it is the one copy
    libvsom_hip.so binds to as well; loaded AFTER libvsom_hip.so has pulled in /opt/rocm's, the process holds two HIP
    runtimes: torch streams / tensors handed to the library belong to the other runtime, and RCCL initialises against
    a runtime that never came up ("no ROCm-capable device is detected" in ncclCommInitAll).  Everything that mixes
    the two -- dist.HipEngine, Group, bench.py -- calls this and fails with the cause instead."""
    libs = hip_runtimes()
    if len(libs) > 1:
        raise VsomError(f"{what}: two HIP runtimes are mapped into this process ({', '.join(libs)}). "
                        "Import torch BEFORE the first vsom_amd call (tests/conftest.py and bench.py do), "
                        "so that libvsom_hip.so binds to the copy torch brings.")


_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VsomError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                        "(variational-self-organizing-maps_amd/csrc/build.sh)")
    L = C.CDLL(LIB_PATH)
    vp, fp, u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)
    L.vsom_last_error.restype = C.c_char_p
    L.vsom_device_count.restype = C.c_int
    L.vsom_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.vsom_destroy.argtypes = [vp]
    L.vsom_destroy.restype = None
    L.vsom_set_stream.argtypes = [vp, vp]
    L.vsom_synchronize.argtypes = [vp]
    L.vsom_set_bmu_mode.argtypes = [vp, C.c_int]
    L.vsom_set_update_mode.argtypes = [vp, C.c_int]
    L.vsom_set_column_compaction.argtypes = [vp, C.c_long]
    L.vsom_set_row_dedupe.argtypes = [vp, C.c_double]
    L.vsom_get_shortlist_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    for name in ("vsom_depth", "vsom_nodes", "vsom_residual_len", "vsom_pitch", "vsom_chunk_pitch"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_uint32
    L.vsom_chunk_size.argtypes = [vp]
    L.vsom_chunk_size.restype = C.c_size_t
    L.vsom_enable_timing_of.argtypes = [vp, C.c_uint32]
    L.vsom_small_map_chains.argtypes = [vp, C.c_size_t]
    L.vsom_small_map_chains.restype = C.c_int
    L.vsom_set_state.argtypes = [vp, fp, fp, fp, fp, u64p]
    L.vsom_get_state.argtypes = [vp, fp, fp, fp, fp, u64p]
    L.vsom_upload_chunk.argtypes = [vp, fp, C.c_size_t]
    L.vsom_upload_chunk_async.argtypes = [vp, fp, C.c_size_t]
    L.vsom_set_chunk_device.argtypes = [vp, vp, C.c_size_t]
    L.vsom_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.vsom_host_free.argtypes = [vp]
    L.vsom_prefetch_chunk.argtypes = [vp, fp, C.c_size_t]
    L.vsom_prefetch_wait.argtypes = [vp]
    L.vsom_commit_chunk.argtypes = [vp]
    L.vsom_stage_next_device.argtypes = [vp, vp, C.c_size_t]
    L.vsom_get_last_bmu.argtypes = [vp, u64p]
    L.vsom_set_last_bmu.argtypes = [vp, u64p]
    L.vsom_get_sqres.argtypes = [vp, fp]
    L.vsom_bmu_batch.argtypes = [vp, u64p, fp]
    L.vsom_find_bmu.argtypes = [vp, fp, u64p, fp]
    L.vsom_bmu_local_batch.argtypes = [vp, u64p, fp]
    L.vsom_distances.argtypes = [vp, u64p, u64p, C.c_size_t, fp]
    L.vsom_bmu_restricted_batch.argtypes = [vp, C.c_uint64, u64p, fp]
    L.vsom_distances_row.argtypes = [vp, C.c_size_t, fp]
    L.vsom_distances_raw.argtypes = [vp, u64p, u64p, C.c_size_t, C.c_int, fp]
    L.vsom_batch_phase1_async.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_int]
    L.vsom_batch_finish_async.argtypes = [vp]
    L.vsom_batch_phase2_async.argtypes = [vp, C.c_double, C.c_size_t, C.c_size_t]
    L.vsom_batch_epoch_async.argtypes = [vp, C.c_double, C.c_int]
    L.vsom_batch_epoch.argtypes = [vp, C.c_double, C.c_int, fp]
    L.vsom_get_mse.argtypes = [vp, fp]
    L.vsom_train_single.argtypes = [vp, fp, C.c_double, C.c_double, u64p, C.c_int, fp, fp, u64p]
    L.vsom_get_online_search_stats.argtypes = [vp, u64p, C.c_int]
    L.vsom_dist_single.argtypes = [vp, fp, C.c_uint64, fp]
    L.vsom_find_local_bmu.argtypes = [vp, fp, C.c_uint64, u64p, fp]
    L.vsom_find_restricted_bmu.argtypes = [vp, fp, C.c_uint64, u64p, fp]
    L.vsom_distances_single.argtypes = [vp, fp, fp]
    L.vsom_train_online_chunk.argtypes = [vp, C.c_double, C.c_double, C.c_int, fp]
    L.vsom_train_online_chunk_acc.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int, fp]
    L.vsom_train_online_chunk_fetch.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_uint64), fp]
    L.vsom_neighbourhood_weight.argtypes = [C.c_size_t] * 4 + [C.c_double]
    L.vsom_neighbourhood_weight.restype = C.c_double
    L.vsom_device_ptr.argtypes = [vp, C.c_int]
    L.vsom_device_ptr.restype = vp
    L.vsom_enable_timing.argtypes = [vp, C.c_int]
    L.vsom_get_timing.argtypes = [vp, fp, C.POINTER(C.c_uint32), C.c_int]
    L.vsom_group_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.vsom_group_destroy.argtypes = [vp]
    L.vsom_group_destroy.restype = None
    L.vsom_group_size.argtypes = [vp]
    L.vsom_group_ctx.argtypes = [vp, C.c_int]
    L.vsom_group_ctx.restype = vp
    L.vsom_group_transport.argtypes = [vp]
    L.vsom_group_transport.restype = C.c_char_p
    L.vsom_group_synchronize.argtypes = [vp]
    L.vsom_group_set_state.argtypes = [vp, fp, fp, fp, fp, u64p]
    L.vsom_group_get_state.argtypes = [vp, fp, fp, fp, fp, u64p]
    L.vsom_group_set_update_mode.argtypes = [vp, C.c_int]
    L.vsom_group_set_bmu_mode.argtypes = [vp, C.c_int]
    L.vsom_group_upload_chunk.argtypes = [vp, fp, C.c_size_t]
    L.vsom_group_prefetch_chunk.argtypes = [vp, fp, C.c_size_t]
    L.vsom_group_prefetch_wait.argtypes = [vp]
    L.vsom_group_commit_chunk.argtypes = [vp]
    L.vsom_group_set_chunk_device.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.vsom_group_set_last_bmu.argtypes = [vp, u64p]
    L.vsom_group_get_last_bmu.argtypes = [vp, u64p]
    L.vsom_group_batch_epoch_async.argtypes = [vp, C.c_double, C.c_int]
    L.vsom_group_batch_epoch.argtypes = [vp, C.c_double, C.c_int, fp]
    L.vsom_group_get_mse.argtypes = [vp, fp]
    _lib = L
    return L


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint64))


def check(rc):
    if rc != 0:
        raise VsomError(f"libvsom_hip error {rc}: {lib().vsom_last_error().decode(errors='replace')}")


def device_count():
    return int(lib().vsom_device_count())


def model_length(transform, in_len):
    """Transformation::Length (Transformation.cpp:33-36,71-74,162-165): J, or J(J-1) for CLR."""
    return int(in_len) * (int(in_len) - 1) if int(transform) == CLR else int(in_len)


def neighbourhood_weight(cx, cy, bx, by, sigma):
    return float(lib().vsom_neighbourhood_weight(cx, cy, bx, by, float(sigma)))


class PinnedBuffer:
    """Pinned host memory (vsom_host_alloc) exposed as a numpy float32 array."""

    def __init__(self, shape):
        self.shape = tuple(int(v) for v in shape)
        n = int(np.prod(self.shape))
        self._p = C.c_void_p()
        check(lib().vsom_host_alloc(C.byref(self._p), max(n, 1) * 4))
        buf = (C.c_float * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.float32, count=n).reshape(self.shape)

    def free(self):
        if self._p:
            self.array = None
            lib().vsom_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """RAII wrapper of a vsom_ctx (one per GPU)."""

    def __init__(self, width, height, in_len, transform=STANDARD, device=0, _borrowed=None):
        self._owned = _borrowed is None
        if _borrowed is None:
            self._h = C.c_void_p()
            check(lib().vsom_create(C.byref(self._h), int(device), int(width), int(height), int(in_len),
                                    int(transform)))
        else:
            self._h = C.c_void_p(_borrowed)       # a member of a Group: the group owns it
        self.width, self.height, self.in_len = int(width), int(height), int(in_len)
        self.transform, self.device = int(transform), int(device)
        self.depth = int(lib().vsom_depth(self._h))
        self.n_nodes = int(lib().vsom_nodes(self._h))
        self.residual_len = int(lib().vsom_residual_len(self._h))
        self.pitch = int(lib().vsom_pitch(self._h))

    def close(self):
        if self._h:
            if self._owned:
                lib().vsom_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing ------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        check(lib().vsom_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def synchronize(self):
        check(lib().vsom_synchronize(self._h))

    def set_bmu_mode(self, mode):
        check(lib().vsom_set_bmu_mode(self._h, int(mode)))

    def set_update_mode(self, mode):
        check(lib().vsom_set_update_mode(self._h, int(mode)))

    def set_row_dedupe(self, min_work):
        """exact searches of at least min_work (sample, node, value) triples evaluate one representative per class of
        bit-identical model rows (default 2e10; 0: always; < 0: never)"""
        check(lib().vsom_set_row_dedupe(self._h, float(min_work)))

    def set_column_compaction(self, min_rows):
        """chunks of at least min_rows rows retire their all-zero columns (default 1024; < 0: off)"""
        check(lib().vsom_set_column_compaction(self._h, int(min_rows)))

    def shortlist_stats(self):
        out = (C.c_uint32 * 4)()
        check(lib().vsom_get_shortlist_stats(self._h, out))
        return {"redo_samples": int(out[0]), "candidates": int(out[1]), "samples": int(out[2]),
                "searches": int(out[3])}

    def device_ptr(self, which):
        return int(lib().vsom_device_ptr(self._h, int(which)) or 0)

    @property
    def chunk_size(self):
        return int(lib().vsom_chunk_size(self._h))

    # ---- state ---------------------------------------------------------
    def set_state(self, map=None, sigma=None, S=None, weight=None, hits=None):
        n, d = self.n_nodes, self.depth

        def f2(a, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            assert a.size == int(np.prod(shape)), (a.shape, shape)
            return a

        m, s, ss, w = f2(map, (n, d)), f2(sigma, (n, d)), f2(S, (n, d)), f2(weight, (n,))
        h = None if hits is None else np.ascontiguousarray(hits, dtype=np.uint64)
        check(lib().vsom_set_state(self._h, _f(m), _f(s), _f(ss), _f(w), _u(h)))

    def get_state(self, map=True, sigma=True, S=True, weight=True, hits=True):
        n, d = self.n_nodes, self.depth
        m = np.empty((n, d), np.float32) if map else None
        s = np.empty((n, d), np.float32) if sigma else None
        ss = np.empty((n, d), np.float32) if S else None
        w = np.empty(n, np.float32) if weight else None
        h = np.empty(n, np.uint64) if hits else None
        check(lib().vsom_get_state(self._h, _f(m), _f(s), _f(ss), _f(w), _u(h)))
        return {"map": m, "sigma": s, "S": ss, "weight": w, "hits": h}

    # ---- chunk ---------------------------------------------------------
    def upload_chunk(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        assert X.ndim == 2 and X.shape[1] == self.in_len, (X.shape, self.in_len)
        check(lib().vsom_upload_chunk(self._h, _f(X), X.shape[0]))

    def upload_chunk_async(self, X_pinned):
        """copy + staging enqueued, no wait: X_pinned (a PinnedBuffer's array) must stay unchanged until a synchronising call"""
        assert X_pinned.ndim == 2 and X_pinned.shape[1] == self.in_len and X_pinned.dtype == np.float32
        check(lib().vsom_upload_chunk_async(self._h, _f(X_pinned), X_pinned.shape[0]))

    def set_chunk_device(self, dev_ptr, B):
        check(lib().vsom_set_chunk_device(self._h, C.c_void_p(int(dev_ptr)), int(B)))

    def prefetch_chunk(self, X):
        """Start the H2D copy of the NEXT chunk (async when X lives in pinned memory, see
        pinned_array); the current chunk keeps training."""
        assert X.dtype == np.float32 and X.flags.c_contiguous and X.ndim == 2 and X.shape[1] == self.in_len
        check(lib().vsom_prefetch_chunk(self._h, _f(X), X.shape[0]))

    def prefetch_wait(self):
        check(lib().vsom_prefetch_wait(self._h))

    def commit_chunk(self):
        check(lib().vsom_commit_chunk(self._h))

    def stage_next_device(self, dev_ptr, n_rows):
        """a next chunk that already lives in HBM: staged beside the running epoch when possible, adopted by
        commit_chunk (the rows must stay valid until then)"""
        check(lib().vsom_stage_next_device(self._h, C.c_void_p(int(dev_ptr) or None), int(n_rows)))

    def get_last_bmu(self):
        out = np.empty(self.chunk_size, np.uint64)
        check(lib().vsom_get_last_bmu(self._h, _u(out)))
        return out

    def set_last_bmu(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        assert idx.size == self.chunk_size
        check(lib().vsom_set_last_bmu(self._h, _u(idx)))

    def get_sqres(self):
        out = np.empty(self.chunk_size, np.float32)
        check(lib().vsom_get_sqres(self._h, _f(out)))
        return out

    # ---- search --------------------------------------------------------
    def bmu_batch(self):
        B = self.chunk_size
        idx, dist = np.empty(B, np.uint64), np.empty(B, np.float32)
        check(lib().vsom_bmu_batch(self._h, _u(idx), _f(dist)))
        return idx, dist

    def dist_single(self, v, node):
        """Som::euclidianWeightedDist(node, v) of one host vector (does not disturb the staged chunk)."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.shape == (self.in_len,)
        d = C.c_float()
        check(lib().vsom_dist_single(self._h, _f(v), int(node), C.byref(d)))
        return np.float32(d.value)

    def find_local_bmu(self, v, last_bmu):
        """Som::findLocalBmu of one host vector from `last_bmu`: (index, distance)."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.shape == (self.in_len,)
        idx, d = C.c_uint64(), C.c_float()
        check(lib().vsom_find_local_bmu(self._h, _f(v), int(last_bmu), C.byref(idx), C.byref(d)))
        return int(idx.value), np.float32(d.value)

    def find_restricted_bmu(self, v, min_hits):
        """Som::findRestrictedBmu of one host vector: (index, distance)."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.shape == (self.in_len,)
        idx, d = C.c_uint64(), C.c_float()
        check(lib().vsom_find_restricted_bmu(self._h, _f(v), int(min_hits), C.byref(idx), C.byref(d)))
        return int(idx.value), np.float32(d.value)

    def distances_single(self, v):
        """euclidianWeightedDist(i, v) of one host vector to every node."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.shape == (self.in_len,)
        out = np.empty(self.n_nodes, np.float32)
        check(lib().vsom_distances_single(self._h, _f(v), _f(out)))
        return out

    def find_bmu(self, v):
        """Som::findBmu of one host vector (does not disturb the staged chunk)."""
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.size == self.in_len
        idx, dist = C.c_uint64(), C.c_float()
        check(lib().vsom_find_bmu(self._h, _f(v), C.byref(idx), C.byref(dist)))
        return int(idx.value), np.float32(dist.value)

    def bmu_local_batch(self):
        B = self.chunk_size
        idx, dist = np.empty(B, np.uint64), np.empty(B, np.float32)
        check(lib().vsom_bmu_local_batch(self._h, _u(idx), _f(dist)))
        return idx, dist

    def distances(self, nodes, rows):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert nodes.size == rows.size
        out = np.empty(nodes.size, np.float32)
        check(lib().vsom_distances(self._h, _u(nodes), _u(rows), nodes.size, _f(out)))
        return out

    def bmu_restricted_batch(self, min_hits):
        B = self.chunk_size
        idx, dist = np.empty(B, np.uint64), np.empty(B, np.float32)
        check(lib().vsom_bmu_restricted_batch(self._h, int(min_hits), _u(idx), _f(dist)))
        return idx, dist

    def distances_row(self, row):
        out = np.empty(self.n_nodes, np.float32)
        check(lib().vsom_distances_row(self._h, int(row), _f(out)))
        return out

    def distances_raw(self, nodes, vrows, from_map):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        vrows = np.ascontiguousarray(vrows, dtype=np.uint64)
        out = np.empty(nodes.size, np.float32)
        check(lib().vsom_distances_raw(self._h, _u(nodes), _u(vrows), nodes.size, int(bool(from_map)), _f(out)))
        return out

    # ---- batch epoch ---------------------------------------------------
    def batch_phase1_async(self, s0, s1, is_first):
        check(lib().vsom_batch_phase1_async(self._h, int(s0), int(s1), int(bool(is_first))))

    def batch_finish_async(self):
        check(lib().vsom_batch_finish_async(self._h))

    def batch_phase2_async(self, sigma, n0, n1):
        check(lib().vsom_batch_phase2_async(self._h, float(sigma), int(n0), int(n1)))

    def batch_epoch_async(self, sigma, is_first):
        check(lib().vsom_batch_epoch_async(self._h, float(sigma), int(bool(is_first))))

    def batch_epoch(self, sigma, is_first):
        mse = C.c_float()
        check(lib().vsom_batch_epoch(self._h, float(sigma), int(bool(is_first)), C.byref(mse)))
        return np.float32(mse.value)

    def get_mse(self):
        mse = C.c_float()
        check(lib().vsom_get_mse(self._h, C.byref(mse)))
        return np.float32(mse.value)

    # ---- online --------------------------------------------------------
    def train_single(self, v, eta, sigma, last_bmu, decay_fn):
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.size == self.in_len
        res = np.empty(self.residual_len, np.float32)
        lb, bmu, dist = C.c_uint64(int(last_bmu)), C.c_uint64(), C.c_float()
        check(lib().vsom_train_single(self._h, _f(v), float(eta), float(sigma), C.byref(lb),
                                      int(decay_fn), _f(res), C.byref(dist), C.byref(bmu)))
        return int(bmu.value), res, np.float32(dist.value), int(lb.value)

    def train_online_chunk(self, eta, sigma, decay_fn, first_chunk=True):
        """B sequential trainSingle steps on the staged chunk; returns the epoch's running MSE
        accumulator after it (first_chunk=False continues the previous chunk's value)."""
        mse = C.c_float()
        check(lib().vsom_train_online_chunk_acc(self._h, float(eta), float(sigma), int(decay_fn),
                                                int(bool(first_chunk)), C.byref(mse)))
        return np.float32(mse.value)

    def train_online_chunk_fetch(self, eta, sigma, decay_fn, first_chunk=True):
        """the same as one synchronising call that also hands back the chunk's lastBMU: (running MSE, lastBMU)"""
        mse = C.c_float()
        lb = np.zeros(self.chunk_size, np.uint64)
        check(lib().vsom_train_online_chunk_fetch(self._h, float(eta), float(sigma), int(decay_fn), int(bool(first_chunk)),
                                                  _u(lb), C.byref(mse)))
        return np.float32(mse.value), lb

    def online_search_stats(self, reset=False):
        """image-bounded search of the online chunk loop: samples searched, nodes evaluated exactly, refinement workgroups
        with work (all zero while the exact scan is in use)"""
        out = np.zeros(4, np.uint64)
        check(lib().vsom_get_online_search_stats(self._h, _u(out), int(bool(reset))))
        return {"samples": int(out[0]), "exact_evaluations": int(out[1]), "refine_workgroups": int(out[2])}

    # ---- measurement ---------------------------------------------------
    def enable_timing(self, on=True, groups=None):
        """HIP-event timing of the kernel groups (TIMER_NAMES); groups = names to time only those (each timed group
        costs two event records between otherwise back-to-back kernels)"""
        if groups is None:
            check(lib().vsom_enable_timing(self._h, int(bool(on))))
        else:
            mask = 0
            for g in groups:
                mask |= 1 << TIMER_NAMES.index(g)
            check(lib().vsom_enable_timing_of(self._h, C.c_uint32(mask if on else 0)))

    def get_timing(self, reset=True):
        ms = (C.c_float * T_COUNT)()
        cnt = (C.c_uint32 * T_COUNT)()
        check(lib().vsom_get_timing(self._h, ms, cnt, int(bool(reset))))
        return {TIMER_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(T_COUNT)}


class Group:
    """RAII wrapper of a vsom_group: Som::trainBatchSomEpoch over several GPUs from one process
    (include/vsom_hip.h, "multi-GPU batch epoch").  devices=None: devices 0..ndev-1; a list that repeats
    a device rehearses the N > 1 flow on one GPU (peer-copy transport)."""

    def __init__(self, width, height, in_len, transform=STANDARD, ndev=0, devices=None):
        self._h = C.c_void_p()
        self._members = []
        self.size = None
        devs = None
        if devices is not None:
            ndev = len(devices)
            devs = (C.c_int * ndev)(*[int(d) for d in devices])
        lib()
        assert_single_hip_runtime("vsom_group_create (RCCL)")
        check(lib().vsom_group_create(C.byref(self._h), int(ndev), devs, int(width), int(height), int(in_len),
                                      int(transform)))
        self.size = int(lib().vsom_group_size(self._h))
        self.transport = lib().vsom_group_transport(self._h).decode()
        self.width, self.height, self.in_len, self.transform = int(width), int(height), int(in_len), int(transform)
        c0 = self.member(0)
        self.depth, self.n_nodes = c0.depth, c0.n_nodes

    def member(self, rank):
        """Borrowed Context of one member for the single-context calls (searches, getters).  The group is
        synchronised first (include/vsom_hip.h requires it), the returned object keeps the group alive, and
        Group.close() invalidates it -- a call on it afterwards raises instead of touching freed memory."""
        if not self._h:
            raise VsomError("group is closed")
        h = lib().vsom_group_ctx(self._h, int(rank))
        if not h:
            raise VsomError("rank out of range")
        if getattr(self, "size", None) is not None:     # (not yet during __init__)
            check(lib().vsom_group_synchronize(self._h))
        c = Context(self.width, self.height, self.in_len, self.transform, _borrowed=h)
        c._group = self
        self._members.append(c)
        return c

    def close(self):
        if self._h:
            for c in self._members:
                c._h = C.c_void_p()                       # borrowed handles die with the group
            self._members = []
            lib().vsom_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().vsom_group_synchronize(self._h))

    def set_update_mode(self, mode):
        check(lib().vsom_group_set_update_mode(self._h, int(mode)))

    def set_bmu_mode(self, mode):
        check(lib().vsom_group_set_bmu_mode(self._h, int(mode)))

    def set_state(self, map=None, sigma=None, S=None, weight=None, hits=None):
        n, d = self.n_nodes, self.depth

        def f2(a, size):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            assert a.size == size, (a.shape, size)
            return a

        m, s, ss, w = f2(map, n * d), f2(sigma, n * d), f2(S, n * d), f2(weight, n)
        h = None if hits is None else np.ascontiguousarray(hits, dtype=np.uint64)
        check(lib().vsom_group_set_state(self._h, _f(m), _f(s), _f(ss), _f(w), _u(h)))

    def get_state(self):
        n, d = self.n_nodes, self.depth
        m, s, ss = (np.empty((n, d), np.float32) for _ in range(3))
        w, h = np.empty(n, np.float32), np.empty(n, np.uint64)
        check(lib().vsom_group_get_state(self._h, _f(m), _f(s), _f(ss), _f(w), _u(h)))
        return {"map": m, "sigma": s, "S": ss, "weight": w, "hits": h}

    def upload_chunk(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        assert X.ndim == 2 and X.shape[1] == self.in_len, (X.shape, self.in_len)
        self._B = X.shape[0]
        check(lib().vsom_group_upload_chunk(self._h, _f(X), X.shape[0]))

    def prefetch_chunk(self, X):
        """X must stay alive (and unchanged) until commit_chunk/prefetch_wait; pinned memory makes it asynchronous"""
        assert X.dtype == np.float32 and X.flags["C_CONTIGUOUS"] and X.shape[1] == self.in_len
        self._B = X.shape[0]
        check(lib().vsom_group_prefetch_chunk(self._h, _f(X), X.shape[0]))

    def prefetch_wait(self):
        check(lib().vsom_group_prefetch_wait(self._h))

    def commit_chunk(self):
        check(lib().vsom_group_commit_chunk(self._h))

    def set_chunk_device(self, rows_dev, B):
        """rows_dev[r] = device pointer (int) of member r's own rows [B*r/n, B*(r+1)/n) on its device"""
        assert len(rows_dev) == self.size
        arr = (C.c_void_p * self.size)(*[C.c_void_p(int(p) or None) for p in rows_dev])
        self._B = int(B)
        check(lib().vsom_group_set_chunk_device(self._h, arr, int(B)))

    def set_last_bmu(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        check(lib().vsom_group_set_last_bmu(self._h, _u(idx)))

    def get_last_bmu(self):
        out = np.empty(self._B, np.uint64)
        check(lib().vsom_group_get_last_bmu(self._h, _u(out)))
        return out

    def batch_epoch_async(self, sigma, is_first):
        check(lib().vsom_group_batch_epoch_async(self._h, float(sigma), int(bool(is_first))))

    def batch_epoch(self, sigma, is_first):
        mse = C.c_float()
        check(lib().vsom_group_batch_epoch(self._h, float(sigma), int(bool(is_first)), C.byref(mse)))
        return np.float32(mse.value)

    def get_mse(self):
        mse = C.c_float()
        check(lib().vsom_group_get_mse(self._h, C.byref(mse)))
        return np.float32(mse.value)
