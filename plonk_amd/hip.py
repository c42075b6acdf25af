"""ctypes binding of libplonk_hip.so (the C ABI declared in include/plonk_hip.h).

Field elements travel as numpy uint64 arrays of shape (n, 4) in Montgomery form (x * 2^256), points as (n, 8)
x||y.  There is no CPU fallback: if the shared library is missing or no GPU is visible every call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLK_LIB_DIR (a directory holding both libraries) and PLK_LIB_OVERRIDE (this library alone) serve kernel experiments (tools/ab.sh)
LIB_DIR = os.environ.get("PLK_LIB_DIR") or os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("PLK_LIB_OVERRIDE") or os.path.join(LIB_DIR, "libplonk_hip.so")

STATUS = {
    0: "PLK_OK", 1: "PLK_ERR_INVALID_ARG", 2: "PLK_ERR_HIP", 3: "PLK_ERR_NO_DEVICE", 4: "PLK_ERR_SRS_TOO_SHORT",
    5: "PLK_ERR_NOT_DIVISIBLE", 6: "PLK_ERR_INVERSE_OF_ZERO", 7: "PLK_ERR_INDEX_OUT_OF_RANGE", 8: "PLK_ERR_ALLOC",
}
# every symbol include/plonk_hip.h declares
EXPORTS = [
    "plk_create", "plk_destroy", "plk_last_error", "plk_malloc", "plk_free", "plk_upload", "plk_upload_enqueue", "plk_host_malloc", "plk_host_free", "plk_download", "plk_sync", "plk_copy_dev",
    "plk_srs_load", "plk_srs_load_shard", "plk_srs_generate", "plk_srs_generate_shard", "plk_srs_generate_cyclic", "plk_srs_load_cyclic", "plk_srs_setup", "plk_srs_setup_cyclic", "plk_srs_read",
    "plk_msm", "plk_msm_dev", "plk_msm_partial_dev", "plk_g1_sum_partials", "plk_msm_plan_query", "plk_ntt", "plk_ntt_dev", "plk_ntt_dev_enqueue",
    "plk_scale_powers_dev", "plk_ntt_cross_dev", "plk_ntt_sharded_dev",
    "plk_interpolate_selector", "plk_interpolate_permutation", "plk_interpolate_permutation_dev",
    "plk_grand_product", "plk_grand_product_dev", "plk_grand_product_values_dev", "plk_grand_product_dev_enqueue", "plk_quotient", "plk_quotient_dev", "plk_quotient_pi", "plk_quotient_pi_dev", "plk_quotient_dev_enqueue", "plk_quotient_dev_finish", "plk_quotient_coset", "plk_quotient_coset_dev", "plk_poly_eval", "plk_poly_eval_dev", "plk_poly_eval_dev_static",
    "plk_kzg_commit", "plk_kzg_commit_dev", "plk_kzg_open", "plk_kzg_open_dev", "plk_kzg_commit_begin", "plk_kzg_commit_begin_batch", "plk_kzg_commit_end",
    "plk_kzg_open_begin", "plk_kzg_open_begin_batch", "plk_kzg_open_end", "plk_kzg_end_partial", "plk_kzg_end_many", "plk_g1_sum_partials_many", "plk_kzg_submit", "plk_kzg_cancel", "plk_kzg_ready", "plk_compute_detach", "plk_comm_get_unique_id", "plk_comm_init", "plk_comm_destroy", "plk_comm_all_gather_partials", "plk_field_op", "plk_profile_enable", "plk_profile_reset",
    "plk_profile_get", "plk_profile_get_busy",
]


class PlonkHipError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        self.status = STATUS.get(code, str(code))
        super().__init__(f"{self.status}: {detail}" if detail else self.status)


_lib = None


def load_library():
    """Loads libplonk_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        # PyTorch wheels carry their own HIP/HSA runtime under the same sonames as /opt/rocm's.  A process can hold only one
        # (the second fails to open the device: "No HIP GPUs are available"), and the dynamic linker shares a copy only when
        # torch's is the one loaded first.  Callers that use torch beside this library (torch.distributed collectives on
        # device buffers) therefore need torch imported before libplonk_hip.so; PLK_NO_TORCH=1 skips this.
        if not os.environ.get("PLK_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C plonk_amd/csrc` (no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH)
        if not os.environ.get("PLK_LIB_DIR"):      # an experiment directory may hold an older build
            for name in EXPORTS:
                getattr(_lib, name)
        _lib.plk_last_error.restype = C.c_char_p
        _lib.plk_last_error.argtypes = [C.c_void_p]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _fr(a, shape=(-1, 4)):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(shape)


def _vp(x):
    return C.c_void_p(x if x else None)


class Context:
    """One plk_ctx on one GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.plk_create(C.c_int32(device), C.byref(h))
        if rc != 0:
            raise PlonkHipError(rc, "plk_create failed (is a GPU visible?)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.plk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PlonkHipError(rc, (self.lib.plk_last_error(self.h) or b"").decode())

    # ---- memory
    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._ck(self.lib.plk_malloc(self.h, C.c_uint64(nbytes), C.byref(p)))
        return p.value

    def free(self, dptr: int):
        self._ck(self.lib.plk_free(self.h, _vp(dptr)))

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        d = self.malloc(max(arr.nbytes, 32))
        if arr.nbytes:
            self._ck(self.lib.plk_upload(self.h, _vp(d), _ptr(arr), C.c_uint64(arr.nbytes)))
        return d

    def upload_into(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._ck(self.lib.plk_upload(self.h, _vp(dptr), _ptr(arr), C.c_uint64(arr.nbytes)))

    def upload_enqueue(self, dptr: int, arr: np.ndarray):
        """plk_upload_enqueue: no wait; arr must stay alive (and unchanged) until results that depend on it have been returned"""
        assert arr.flags["C_CONTIGUOUS"]
        self._ck(self.lib.plk_upload_enqueue(self.h, _vp(dptr), _ptr(arr), C.c_uint64(arr.nbytes)))

    def host_empty(self, shape, dtype=np.uint64) -> np.ndarray:
        """numpy array in page-locked host memory (plk_host_malloc); freed with host_free(arr)"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._ck(self.lib.plk_host_malloc(self.h, C.c_uint64(nbytes), C.byref(p)))
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.__array_interface__["data"][0]] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = self._pinned.pop(arr.__array_interface__["data"][0])
        self._ck(self.lib.plk_host_free(self.h, _vp(p)))

    def download(self, dptr: int, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            self._ck(self.lib.plk_download(self.h, _ptr(out), _vp(dptr), C.c_uint64(out.nbytes)))
        return out

    # ---- RCCL inside the library
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = self.lib.plk_comm_get_unique_id(buf)
        if rc != 0:
            raise PlonkHipError(rc, "plk_comm_get_unique_id failed (librccl not found?)")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self.lib.plk_comm_init(self.h, buf, C.c_uint32(rank), C.c_uint32(world)))

    def comm_destroy(self):
        self._ck(self.lib.plk_comm_destroy(self.h))

    def comm_all_gather_partials(self, mine: np.ndarray, world: int) -> np.ndarray:
        m = np.ascontiguousarray(mine, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros((world, len(m), 16), dtype=np.uint64)
        self._ck(self.lib.plk_comm_all_gather_partials(self.h, _ptr(m), C.c_uint64(len(m)), _ptr(out)))
        return out

    def compute_detach(self, on):
        """0 re-attach and order, 1 detach, 2 re-attach without ordering (include/plonk_hip.h)"""
        self._ck(self.lib.plk_compute_detach(self.h, C.c_int32(int(on))))

    def quotient_dev_enqueue(self, d_a, d_b, d_c, d_qa, d_qm, d_zv, d_sigma, log_n, beta, gamma, d_t):
        self._ck(self.lib.plk_quotient_dev_enqueue(self.h, _vp(d_a), _vp(d_b), _vp(d_c), _vp(d_qa), _vp(d_qm), _vp(d_zv), _vp(d_sigma), C.c_uint32(log_n),
                                                   _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))), _vp(d_t)))

    def quotient_dev_finish(self):
        tlen = C.c_uint64(0)
        self._ck(self.lib.plk_quotient_dev_finish(self.h, C.byref(tlen)))
        return tlen.value

    def sync(self):
        self._ck(self.lib.plk_sync(self.h))

    # ---- SRS
    def srs_load(self, g_xy, gamma_g_xy, shard_first=0, total_len=None):
        g = _fr(g_xy, (-1, 8))
        gg = _fr(gamma_g_xy, (-1, 8)) if gamma_g_xy is not None else None
        total = len(g) if total_len is None else total_len
        self._ck(self.lib.plk_srs_load_shard(self.h, _ptr(g), C.c_uint64(shard_first), C.c_uint64(len(g)), C.c_uint64(total), _ptr(gg)))

    def srs_generate(self, tau, gscal, length, shard_first=0, shard_len=None):
        sl = length if shard_len is None else shard_len
        self._ck(self.lib.plk_srs_generate_shard(self.h, _ptr(_fr(tau, (4,))), _ptr(_fr(gscal, (4,))), C.c_uint64(shard_first), C.c_uint64(sl),
                                                 C.c_uint64(length)))

    def srs_generate_cyclic(self, tau, gscal, length, rank, world, log_block=12):
        self._ck(self.lib.plk_srs_generate_cyclic(self.h, _ptr(_fr(tau, (4,))), _ptr(_fr(gscal, (4,))), C.c_uint32(rank), C.c_uint32(world),
                                                  C.c_uint32(log_block), C.c_uint64(length)))

    def srs_setup(self, beta, g_xy, gamma_g_xy, length, rank=0, world=1, log_block=12):
        """powers_of_g[i] = beta^i g, powers_of_gamma_g[j] = beta^j gamma_g for arbitrary points g, gamma_g (PCS::setup, src/main.rs:182)"""
        self._ck(self.lib.plk_srs_setup_cyclic(self.h, _ptr(_fr(beta, (4,))), _ptr(_fr(g_xy, (8,))), _ptr(_fr(gamma_g_xy, (8,))), C.c_uint32(rank),
                                               C.c_uint32(world), C.c_uint32(log_block), C.c_uint64(length)))

    def srs_load_cyclic(self, g_xy, gamma_g_xy, rank, world, log_block=12):
        g = _fr(g_xy, (-1, 8))
        gg = _fr(gamma_g_xy, (-1, 8)) if gamma_g_xy is not None else None
        self._ck(self.lib.plk_srs_load_cyclic(self.h, _ptr(g), C.c_uint64(len(g)), _ptr(gg), C.c_uint32(rank), C.c_uint32(world), C.c_uint32(log_block)))

    def srs_read(self, which, first, count):
        out = np.empty((count, 8), dtype=np.uint64)
        self._ck(self.lib.plk_srs_read(self.h, C.c_int32(which), C.c_uint64(first), C.c_uint64(count), _ptr(out)))
        return out

    # ---- MSM
    def msm(self, scalars, base_offset=0):
        s = _fr(scalars)
        out = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        self._ck(self.lib.plk_msm(self.h, _ptr(s) if len(s) else None, C.c_uint64(len(s)), C.c_uint64(base_offset), _ptr(out), C.byref(inf)))
        return out, inf.value

    def msm_dev(self, d_scalars, k, base_offset=0):
        out = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        self._ck(self.lib.plk_msm_dev(self.h, _vp(d_scalars), C.c_uint64(k), C.c_uint64(base_offset), _ptr(out), C.byref(inf)))
        return out, inf.value

    def msm_partial_dev(self, d_scalars, k, base_offset=0):
        out = np.empty(16, dtype=np.uint64)
        self._ck(self.lib.plk_msm_partial_dev(self.h, _vp(d_scalars), C.c_uint64(k), C.c_uint64(base_offset), _ptr(out)))
        return out

    def msm_plan(self, k, base_offset=0):
        """(bits per window, windows per scalar, buckets) of the MSM over [base_offset, base_offset + k) on this context"""
        c, w, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        if os.environ.get("PLK_LIB_DIR") and not hasattr(self.lib, "plk_msm_plan_query"):
            return 16, 16, 1 << 15     # an experiment directory with a build from before the query existed (16-bit windows throughout then)
        self._ck(self.lib.plk_msm_plan_query(self.h, C.c_uint64(k), C.c_uint64(base_offset), C.byref(c), C.byref(w), C.byref(b)))
        return c.value, w.value, b.value

    def sum_partials(self, parts):
        p = _fr(parts, (-1, 16))
        out = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        self._ck(self.lib.plk_g1_sum_partials(_ptr(p), C.c_uint64(len(p)), _ptr(out), C.byref(inf)))
        return out, inf.value

    # ---- NTT
    def ntt(self, data, inverse=False, coset=None):
        a = _fr(data).copy()
        log_n = (len(a) - 1).bit_length()
        assert 1 << log_n == len(a)
        self._ck(self.lib.plk_ntt(self.h, _ptr(a), C.c_uint32(log_n), C.c_int32(int(inverse)), _ptr(_fr(coset, (4,))) if coset is not None else None))
        return a

    def ntt_dev(self, dptr, log_n, inverse=False, coset=None):
        self._ck(self.lib.plk_ntt_dev(self.h, _vp(dptr), C.c_uint32(log_n), C.c_int32(int(inverse)),
                                      _ptr(_fr(coset, (4,))) if coset is not None else None))

    def ntt_dev_enqueue(self, dptr, log_n, inverse=False, coset=None):
        self._ck(self.lib.plk_ntt_dev_enqueue(self.h, _vp(dptr), C.c_uint32(log_n), C.c_int32(int(inverse)),
                                              _ptr(_fr(coset, (4,))) if coset is not None else None))

    def scale_powers_dev(self, dptr, n, base, ratio):
        self._ck(self.lib.plk_scale_powers_dev(self.h, _vp(dptr), C.c_uint64(n), _ptr(_fr(base, (4,))), _ptr(_fr(ratio, (4,)))))

    def ntt_cross_dev(self, dptr, log_g, cols, inverse=False):
        self._ck(self.lib.plk_ntt_cross_dev(self.h, _vp(dptr), C.c_uint32(log_g), C.c_uint64(cols), C.c_int32(int(inverse))))

    def ntt_sharded_dev(self, d_data, d_tmp, log_n, inverse=False, coset=None, phase=0, rank=0, world=1) -> bool:
        """plk_ntt_sharded_dev; returns True when the result is in d_tmp"""
        r = C.c_int32(0)
        self._ck(self.lib.plk_ntt_sharded_dev(self.h, _vp(d_data), _vp(d_tmp), C.c_uint32(log_n), C.c_int32(int(inverse)),
                                              _ptr(_fr(coset, (4,))) if coset is not None else None, C.c_uint32(phase), C.c_uint32(rank), C.c_uint32(world), C.byref(r)))
        return bool(r.value)

    def interpolate_selector(self, values):
        v = _fr(values)
        n = 1
        while n < len(v):
            n *= 2
        out = np.zeros((n, 4), dtype=np.uint64); log_n = C.c_uint32(0)
        self._ck(self.lib.plk_interpolate_selector(self.h, _ptr(v), C.c_uint64(len(v)), _ptr(out), C.byref(log_n)))
        return out

    def interpolate_permutation(self, sigma, log_n4):
        s = np.ascontiguousarray(sigma, dtype=np.uint64)
        n4 = 1 << log_n4
        sid = np.empty((n4, 4), dtype=np.uint64); ssg = np.empty((n4, 4), dtype=np.uint64)
        self._ck(self.lib.plk_interpolate_permutation(self.h, _ptr(s), C.c_uint64(len(s)), C.c_uint32(log_n4), _ptr(sid), _ptr(ssg)))
        return sid, ssg

    def interpolate_permutation_dev(self, d_sigma, sigma_len, log_n4, d_sid, d_ssigma):
        self._ck(self.lib.plk_interpolate_permutation_dev(self.h, _vp(d_sigma), C.c_uint64(sigma_len), C.c_uint32(log_n4), _vp(d_sid), _vp(d_ssigma)))

    # ---- rounds
    def grand_product(self, witness_flat, sigma, beta, gamma):
        w = _fr(witness_flat); s = np.ascontiguousarray(sigma, dtype=np.uint64)
        n = len(w) // 3; log_n = (n - 1).bit_length()
        out = np.empty((n, 4), dtype=np.uint64)
        self._ck(self.lib.plk_grand_product(self.h, _ptr(w), _ptr(s), C.c_uint32(log_n), _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))), _ptr(out)))
        return out

    def grand_product_dev(self, d_w, d_sigma, log_n, beta, gamma, d_out):
        self._ck(self.lib.plk_grand_product_dev(self.h, _vp(d_w), _vp(d_sigma), C.c_uint32(log_n), _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))), _vp(d_out)))

    def grand_product_values_dev(self, d_w, d_sigma, log_n, beta, gamma, d_vals, d_out):
        self._ck(self.lib.plk_grand_product_values_dev(self.h, _vp(d_w), _vp(d_sigma), C.c_uint32(log_n), _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))),
                                                       _vp(d_vals), _vp(d_out)))

    def grand_product_dev_enqueue(self, d_w, d_sigma, log_n, beta, gamma, d_vals, d_out):
        """grand_product_values_dev without the final wait (sync() before reading the outputs on the host)"""
        self._ck(self.lib.plk_grand_product_dev_enqueue(self.h, _vp(d_w), _vp(d_sigma), C.c_uint32(log_n), _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))),
                                                        _vp(d_vals), _vp(d_out)))

    def ready(self, ticket) -> bool:
        r = C.c_int32(0)
        self._ck(self.lib.plk_kzg_ready(self.h, C.c_int32(ticket), C.byref(r)))
        return bool(r.value)

    def cancel(self):
        """gives up every open ticket (after a failed call in the middle of a batch)"""
        self._ck(self.lib.plk_kzg_cancel(self.h))

    def submit(self):
        self._ck(self.lib.plk_kzg_submit(self.h))

    def quotient(self, a, b, c, qa, qm, zv, sigma, beta, gamma):
        a = _fr(a); n = len(a); log_n = (n - 1).bit_length()
        out = np.empty((n, 4), dtype=np.uint64); tlen = C.c_uint64(0)
        self._ck(self.lib.plk_quotient(self.h, _ptr(a), _ptr(_fr(b)), _ptr(_fr(c)), _ptr(_fr(qa)), _ptr(_fr(qm)), _ptr(_fr(zv)),
                                       _ptr(np.ascontiguousarray(sigma, dtype=np.uint64)), C.c_uint32(log_n), _ptr(_fr(beta, (4,))),
                                       _ptr(_fr(gamma, (4,))), _ptr(out), C.byref(tlen)))
        return out, tlen.value

    def quotient_pi(self, a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, pi_rows, pi_values):
        """plk_quotient_pi: build_quotient_polynomial with public inputs (rows and values, src/circuit.rs:410-439)"""
        a = _fr(a); n = len(a); log_n = (n - 1).bit_length()
        rows = np.ascontiguousarray(pi_rows, dtype=np.uint64); vals = _fr(pi_values)
        out = np.empty((n, 4), dtype=np.uint64); tlen = C.c_uint64(0)
        self._ck(self.lib.plk_quotient_pi(self.h, _ptr(a), _ptr(_fr(b)), _ptr(_fr(c)), _ptr(_fr(qa)), _ptr(_fr(qm)), _ptr(_fr(zv)),
                                          _ptr(np.ascontiguousarray(sigma, dtype=np.uint64)), C.c_uint32(log_n), _ptr(_fr(beta, (4,))),
                                          _ptr(_fr(gamma, (4,))), _ptr(_fr(alpha, (4,))), _ptr(rows), _ptr(vals), C.c_uint64(len(rows)), _ptr(out), C.byref(tlen)))
        return out, tlen.value

    def quotient_dev(self, d_a, d_b, d_c, d_qa, d_qm, d_zv, d_sigma, log_n, beta, gamma, d_t):
        tlen = C.c_uint64(0)
        self._ck(self.lib.plk_quotient_dev(self.h, _vp(d_a), _vp(d_b), _vp(d_c), _vp(d_qa), _vp(d_qm), _vp(d_zv), _vp(d_sigma), C.c_uint32(log_n),
                                           _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))), _vp(d_t), C.byref(tlen)))
        return tlen.value

    def quotient_coset(self, a, b, c, qa, qm, zv, sigma, beta, gamma, alpha, coset=None):
        """4n coefficients of the extended-coset quotient; raises PLK_ERR_NOT_DIVISIBLE for a witness that does not satisfy the constraints."""
        a = _fr(a); n = len(a); log_n = (n - 1).bit_length()
        out = np.empty((4 * n, 4), dtype=np.uint64)
        self._ck(self.lib.plk_quotient_coset(self.h, _ptr(a), _ptr(_fr(b)), _ptr(_fr(c)), _ptr(_fr(qa)), _ptr(_fr(qm)), _ptr(_fr(zv)),
                                             _ptr(np.ascontiguousarray(sigma, dtype=np.uint64)), C.c_uint32(log_n), _ptr(_fr(beta, (4,))),
                                             _ptr(_fr(gamma, (4,))), _ptr(_fr(alpha, (4,))), _ptr(_fr(coset, (4,))) if coset is not None else None, _ptr(out)))
        return out

    def quotient_coset_dev(self, d_a, d_b, d_c, d_qa, d_qm, d_zv, d_sigma, log_n, beta, gamma, alpha, d_t, coset=None):
        self._ck(self.lib.plk_quotient_coset_dev(self.h, _vp(d_a), _vp(d_b), _vp(d_c), _vp(d_qa), _vp(d_qm), _vp(d_zv), _vp(d_sigma), C.c_uint32(log_n),
                                                 _ptr(_fr(beta, (4,))), _ptr(_fr(gamma, (4,))), _ptr(_fr(alpha, (4,))),
                                                 _ptr(_fr(coset, (4,))) if coset is not None else None, _vp(d_t)))

    def poly_eval(self, coeffs, x):
        c = _fr(coeffs); out = np.empty(4, dtype=np.uint64)
        self._ck(self.lib.plk_poly_eval(self.h, _ptr(c) if len(c) else None, C.c_uint64(len(c)), _ptr(_fr(x, (4,))), _ptr(out)))
        return out

    def poly_eval_dev_static(self, d_coeffs, length, x, d_out):
        """plk_poly_eval_dev_static: the value goes to the 32 bytes of device memory at d_out; enqueued, not waited for"""
        self._ck(self.lib.plk_poly_eval_dev_static(self.h, _vp(d_coeffs), C.c_uint64(length), _ptr(_fr(x, (4,))), _vp(d_out)))

    def poly_eval_dev(self, d_coeffs, length, x):
        out = np.empty(4, dtype=np.uint64)
        self._ck(self.lib.plk_poly_eval_dev(self.h, _vp(d_coeffs), C.c_uint64(length), _ptr(_fr(x, (4,))), _ptr(out)))
        return out

    # ---- KZG
    def kzg_commit(self, coeffs, blind):
        c = _fr(coeffs); out = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        self._ck(self.lib.plk_kzg_commit(self.h, _ptr(c) if len(c) else None, C.c_uint64(len(c)), _ptr(_fr(blind, (12,))), _ptr(out), C.byref(inf)))
        return out, inf.value

    def kzg_open(self, coeffs, blind, zeta, chal):
        c = _fr(coeffs); w = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        rv = np.empty(4, dtype=np.uint64); ev = np.empty(4, dtype=np.uint64)
        self._ck(self.lib.plk_kzg_open(self.h, _ptr(c) if len(c) else None, C.c_uint64(len(c)), _ptr(_fr(blind, (12,))), _ptr(_fr(zeta, (4,))),
                                       _ptr(_fr(chal, (4,))), _ptr(w), C.byref(inf), _ptr(rv), _ptr(ev)))
        return w, inf.value, rv, ev

    def commit_begin(self, d_coeffs, length, blind) -> int:
        t = C.c_int32(-1)
        self._ck(self.lib.plk_kzg_commit_begin(self.h, _vp(d_coeffs), C.c_uint64(length), _ptr(_fr(blind, (12,))), C.byref(t)))
        return t.value

    def commit_end(self, ticket):
        out = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        self._ck(self.lib.plk_kzg_commit_end(self.h, C.c_int32(ticket), _ptr(out), C.byref(inf)))
        return out, inf.value

    def open_begin(self, d_coeffs, length, blind, zeta, chal) -> int:
        t = C.c_int32(-1)
        self._ck(self.lib.plk_kzg_open_begin(self.h, _vp(d_coeffs), C.c_uint64(length), _ptr(_fr(blind, (12,))), _ptr(_fr(zeta, (4,))),
                                             _ptr(_fr(chal, (4,))), C.byref(t)))
        return t.value

    def commit_begin_batch(self, d_coeffs, lengths, blinds):
        """plk_kzg_commit_begin_batch -> list of tickets"""
        count = len(d_coeffs)
        ptrs = (C.c_void_p * count)(*[C.c_void_p(p if p else None) for p in d_coeffs])
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        b = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(count, 12)
        t = np.full(count, -1, dtype=np.int32)
        self._ck(self.lib.plk_kzg_commit_begin_batch(self.h, C.c_uint32(count), ptrs, _ptr(lens), _ptr(b), _ptr(t)))
        return [int(x) for x in t]

    def open_begin_batch(self, d_coeffs, lengths, blinds, zeta, chal):
        """plk_kzg_open_begin_batch: d_coeffs / lengths per polynomial, blinds (count, 12) -> list of tickets"""
        count = len(d_coeffs)
        ptrs = (C.c_void_p * count)(*[C.c_void_p(p if p else None) for p in d_coeffs])
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        b = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(count, 12)
        t = np.full(count, -1, dtype=np.int32)
        self._ck(self.lib.plk_kzg_open_begin_batch(self.h, C.c_uint32(count), ptrs, _ptr(lens), _ptr(b), _ptr(_fr(zeta, (4,))), _ptr(_fr(chal, (4,))), _ptr(t)))
        return [int(x) for x in t]

    def open_end(self, ticket):
        w = np.empty(8, dtype=np.uint64); inf = C.c_int32(0)
        rv = np.empty(4, dtype=np.uint64); ev = np.empty(4, dtype=np.uint64)
        self._ck(self.lib.plk_kzg_open_end(self.h, C.c_int32(ticket), _ptr(w), C.byref(inf), _ptr(rv), _ptr(ev)))
        return w, inf.value, rv, ev

    def end_many(self, tickets, affine=True):
        """plk_kzg_end_many -> (points (count, 16), is_inf (count,), random_v (count, 4), eval (count, 4))"""
        t = np.ascontiguousarray(tickets, dtype=np.int32)
        n = len(t)
        out = np.zeros((n, 16), dtype=np.uint64); inf = np.zeros(n, dtype=np.int32)
        rv = np.zeros((n, 4), dtype=np.uint64); ev = np.zeros((n, 4), dtype=np.uint64)
        self._ck(self.lib.plk_kzg_end_many(self.h, C.c_uint32(n), _ptr(t), C.c_int32(int(affine)), _ptr(out), _ptr(inf), _ptr(rv), _ptr(ev)))
        return out, inf, rv, ev

    def sum_partials_many(self, parts):
        """parts: (world, count, 16) -> (xy (count, 8), is_inf (count,))"""
        p = np.ascontiguousarray(parts, dtype=np.uint64)
        world, count = p.shape[0], p.shape[1]
        out = np.zeros((count, 8), dtype=np.uint64); inf = np.zeros(count, dtype=np.int32)
        self._ck(self.lib.plk_g1_sum_partials_many(_ptr(p), C.c_uint64(world), C.c_uint64(count), _ptr(out), _ptr(inf)))
        return out, inf

    def end_partial(self, ticket):
        out = np.empty(16, dtype=np.uint64); rv = np.zeros(4, dtype=np.uint64); ev = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.plk_kzg_end_partial(self.h, C.c_int32(ticket), _ptr(out), _ptr(rv), _ptr(ev)))
        return out, rv, ev

    # ---- misc
    def field_op(self, field, op, a, b):
        a = _fr(a); b = _fr(b); out = np.empty_like(a)
        self._ck(self.lib.plk_field_op(self.h, C.c_int32({"fr": 0, "fq": 1}[field]), C.c_int32({"add": 0, "sub": 1, "mul": 2}[op]), _ptr(a), _ptr(b),
                                       _ptr(out), C.c_uint64(len(a))))
        return out

    def profile_enable(self, on=True):
        self._ck(self.lib.plk_profile_enable(self.h, C.c_int32(int(on))))

    def profile_reset(self):
        self._ck(self.lib.plk_profile_reset(self.h))

    def profile_get(self, name):
        ms = C.c_double(0); n = C.c_uint64(0)
        self._ck(self.lib.plk_profile_get(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_get_busy(self, name):
        ms = C.c_double(0)
        self._ck(self.lib.plk_profile_get_busy(self.h, name.encode(), C.byref(ms)))
        return ms.value
