"""ctypes mirror of the reference's operator interface for the hot path, over the C ABI of libbbgpu.so.

Names follow the reference: ``pippenger`` / ``batched_scalar_multiplications``
(src/barretenberg/curves/bn254/scalar_multiplication.hpp:60-96) and ``fft`` / ``ifft`` / ``coset_fft`` / ``coset_ifft`` /
``fft_with_constant`` / ``ifft_with_constant`` / ``coset_fft_with_constant``
(src/barretenberg/polynomials/polynomial_arithmetic.hpp:27-41).  Arrays are numpy uint64 in the reference's memory
layout: field element (4,), affine point (8,), Jacobian (12,).  Device-resident variants take raw device pointers
(e.g. ``torch.Tensor.data_ptr()``); torch is plumbing only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

try:  # plumbing: when torch is around, let it load ITS HIP runtime first so that device pointers handed over from torch
    import torch  # noqa: F401  tensors and libbbgpu.so share one runtime (two runtimes in one process cannot both see the GPU)
except Exception:  # pragma: no cover - torch is optional for the library itself
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of two streams that share a queue run one
# after the other.  The library's MSM slots want their own queues (libbbgpu sets the same default when it makes the process's first HIP call);
# effective only if HIP has not been initialised yet -- import this module before torch touches the GPU, or export the variable.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
u64p = C.POINTER(C.c_uint64)

NTT_KINDS = {
    "fft": 0,
    "ifft": 1,
    "coset_fft": 2,
    "coset_ifft": 3,
    "fft_with_constant": 4,
    "ifft_with_constant": 5,
    "coset_fft_with_constant": 6,
}

C_ABI_SYMBOLS = [
    "bbgpu_init", "bbgpu_shutdown", "bbgpu_device_count", "bbgpu_last_error", "bbgpu_version", "bbgpu_ntt",
    "bbgpu_ntt_device", "bbgpu_ntt_device_batch", "bbgpu_srs_register", "bbgpu_srs_release", "bbgpu_srs_generate", "bbgpu_srs_generate_range", "bbgpu_set_precompute",
    "bbgpu_srs_num_windows", "bbgpu_transcript_read_g1", "bbgpu_transcript_write", "bbgpu_msm_g1", "bbgpu_msm_g1_plain",
    "bbgpu_msm_g1_batch", "bbgpu_msm_num_windows", "bbgpu_msm_g1_device", "bbgpu_msm_g1_device_async", "bbgpu_msm_g1_device_rows_async", "bbgpu_msm_g1_device_buckets_async", "bbgpu_srs_has_window_tables", "bbgpu_msm_g1_wait",
    "bbgpu_msm_g1_device_batch_async", "bbgpu_msm_g1_batch_wait",
    "bbgpu_g1_sum", "bbgpu_last_timing", "bbgpu_set_host_thresholds", "bbgpu_srs_cache_stats", "bbgpu_set_table_share", "bbgpu_set_point_share", "bbgpu_selftest_field", "bbgpu_selftest_g1",
    "bbgpu_set_timing",
    "bbgpu_fr_evaluate_device", "bbgpu_fr_batch_invert_device", "bbgpu_fr_product_scan_device", "bbgpu_fr_mul_device",
    "bbgpu_kate_opening_device", "bbgpu_lagrange_l1_fft_device", "bbgpu_divide_by_pseudo_vanishing_device",
    "bbgpu_permutation_lagrange_base_device",
    "bbgpu_fr_evaluate", "bbgpu_kate_opening", "bbgpu_lagrange_l1_fft", "bbgpu_divide_by_pseudo_vanishing", "bbgpu_lagrange_evaluations",
    "bbgpu_generate_point_table",
    "bbgpu_plonk_prover_create", "bbgpu_plonk_prover_set_witness", "bbgpu_plonk_construct_proof", "bbgpu_plonk_preprocess", "bbgpu_plonk_last_challenges",
    "bbgpu_plonk_last_timing", "bbgpu_plonk_prover_destroy", "bbgpu_plonk_challenges_from_proof",
    "bbgpu_host_msm_g1", "bbgpu_host_ntt", "bbgpu_host_fr_evaluate", "bbgpu_host_kate_opening", "bbgpu_host_lagrange_l1_fft",
    "bbgpu_host_divide_by_pseudo_vanishing", "bbgpu_memory_stats", "bbgpu_fault_inject", "bbgpu_fault_stats", "bbgpu_srs_set_validate",
]


class MemoryInfo(C.Structure):
    """bbgpu_memory_info (include/bbgpu.h)"""
    _fields_ = [(k, C.c_uint64) for k in ("srs_points_bytes", "srs_table_bytes", "srs_auto_bytes", "srs_cache_cap_bytes", "ntt_table_bytes", "ntt_table_cap_bytes",
                                          "ntt_table_sets", "msm_workspace_bytes", "staging_bytes", "pinned_host_bytes")]


class FaultInfo(C.Structure):
    """bbgpu_fault_info (include/bbgpu.h)"""
    _fields_ = [(k, C.c_uint64) for k in ("alloc_calls", "h2d_calls", "d2h_calls", "launch_checks", "armed", "fired", "absorbed", "live_allocations",
                                          "live_bytes", "slots_pending")]


class BbGpuError(RuntimeError):
    pass


def library_path():
    # BBGPU_LIB: alternative build of the same library (A/B timing of kernel variants); default = the in-tree build
    return os.environ.get("BBGPU_LIB") or os.path.join(HERE, "libbbgpu.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of libbbgpu.so, in-tree (cross-compiles without a GPU)."""
    src = os.path.join(HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", src, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", src, "-j4"], check=True, stdout=subprocess.DEVNULL)
    return library_path()


class MsmJob(C.Structure):
    """layout of scalar_multiplication::multiplication_state (scalar_multiplication.hpp:88-94)"""
    _fields_ = [("points", u64p), ("scalars", u64p), ("num_elements", C.c_size_t), ("_pad", C.c_uint64),
                ("output", C.c_uint64 * 12)]


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need C-contiguous uint64"
    return a.ctypes.data_as(u64p)


class BbGpu:
    def __init__(self, device=0, init=True):
        path = library_path()
        if not os.path.exists(path):
            raise BbGpuError("libbbgpu.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = self.lib = C.CDLL(path)
        L.bbgpu_last_error.restype = C.c_char_p
        L.bbgpu_version.restype = C.c_char_p
        L.bbgpu_ntt.argtypes = [u64p, C.c_size_t, C.c_int, u64p]
        L.bbgpu_ntt_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, u64p, C.c_void_p]
        L.bbgpu_ntt_device_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, u64p, C.c_void_p]
        L.bbgpu_srs_register.argtypes = [u64p, C.c_size_t]
        L.bbgpu_srs_generate.argtypes = [u64p, C.c_size_t, u64p]
        if hasattr(L, "bbgpu_srs_generate_range"):  # absent from older A/B builds loaded through BBGPU_LIB
            L.bbgpu_srs_generate_range.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p]
        L.bbgpu_msm_g1.argtypes = [u64p, u64p, C.c_size_t, u64p]
        if hasattr(L, "bbgpu_fault_inject"):  # absent from older A/B builds loaded through BBGPU_LIB
            L.bbgpu_fault_inject.argtypes = [C.c_char_p]
        L.bbgpu_msm_g1_plain.argtypes = [u64p, u64p, C.c_size_t, u64p]
        L.bbgpu_msm_g1_batch.argtypes = [C.POINTER(MsmJob), C.c_size_t]
        L.bbgpu_msm_num_windows.argtypes = [C.c_size_t]
        L.bbgpu_srs_num_windows.argtypes = [C.c_int, C.c_size_t]
        L.bbgpu_msm_g1_device.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, u64p, C.c_void_p]
        L.bbgpu_msm_g1_device_async.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.bbgpu_msm_g1_device_rows_async.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]
        L.bbgpu_srs_has_window_tables.argtypes = [C.c_int]
        L.bbgpu_msm_g1_wait.argtypes = [C.c_int, u64p]
        L.bbgpu_msm_g1_device_batch_async.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_void_p]
        L.bbgpu_msm_g1_batch_wait.argtypes = [C.c_int, u64p]
        L.bbgpu_g1_sum.argtypes = [u64p, C.c_size_t, u64p]
        L.bbgpu_last_timing.argtypes = [C.POINTER(C.c_float), C.c_int]
        vp, sz = C.c_void_p, C.c_size_t
        L.bbgpu_fr_evaluate_device.argtypes = [vp, sz, u64p, u64p, vp]
        L.bbgpu_fr_batch_invert_device.argtypes = [vp, sz, vp]
        L.bbgpu_fr_product_scan_device.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp]
        L.bbgpu_fr_mul_device.argtypes = [vp, vp, vp, sz, vp]
        L.bbgpu_kate_opening_device.argtypes = [vp, vp, sz, u64p, u64p, vp]
        L.bbgpu_lagrange_l1_fft_device.argtypes = [vp, sz, sz, vp]
        L.bbgpu_divide_by_pseudo_vanishing_device.argtypes = [vp, sz, sz, vp]
        L.bbgpu_permutation_lagrange_base_device.argtypes = [vp, vp, sz, vp]
        self.device = device
        if init:
            self._chk(L.bbgpu_init(device))

    def _chk(self, rc):
        if rc < 0:
            raise BbGpuError("bbgpu error %d: %s" % (rc, self.lib.bbgpu_last_error().decode()))
        return rc

    def version(self):
        return self.lib.bbgpu_version().decode()

    def device_count(self):
        return int(self.lib.bbgpu_device_count())

    def shutdown(self):
        self.lib.bbgpu_shutdown()

    # ---- polynomial_arithmetic ------------------------------------------------------------------------------------
    def ntt(self, coeffs, kind, constant=None):
        """in place on a (n, 4) uint64 host array"""
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        cp = _ptr(np.ascontiguousarray(constant, dtype=np.uint64)) if constant is not None else None
        self._chk(self.lib.bbgpu_ntt(_ptr(coeffs), coeffs.shape[0], kind, cp))
        return coeffs

    def fft(self, coeffs): return self.ntt(coeffs, "fft")
    def ifft(self, coeffs): return self.ntt(coeffs, "ifft")
    def coset_fft(self, coeffs): return self.ntt(coeffs, "coset_fft")
    def coset_ifft(self, coeffs): return self.ntt(coeffs, "coset_ifft")
    def fft_with_constant(self, coeffs, value): return self.ntt(coeffs, "fft_with_constant", value)
    def ifft_with_constant(self, coeffs, value): return self.ntt(coeffs, "ifft_with_constant", value)
    def coset_fft_with_constant(self, coeffs, constant): return self.ntt(coeffs, "coset_fft_with_constant", constant)

    def ntt_device(self, d_ptr, n, kind, constant=None, stream=None):
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        cp = _ptr(np.ascontiguousarray(constant, dtype=np.uint64)) if constant is not None else None
        self._chk(self.lib.bbgpu_ntt_device(C.c_void_p(d_ptr), n, kind, cp, C.c_void_p(stream or 0)))

    def ntt_device_batch(self, d_ptr, n, batch, kind, constant=None, stride=None, stream=None):
        """`batch` transforms of consecutive n-element vectors (stride elements apart) in one set of launches"""
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        cp = _ptr(np.ascontiguousarray(constant, dtype=np.uint64)) if constant is not None else None
        self._chk(self.lib.bbgpu_ntt_device_batch(C.c_void_p(d_ptr), n, stride or n, batch, kind, cp, C.c_void_p(stream or 0)))

    # ---- the O(n) helpers between transforms and commitments, on device-resident vectors (raw device pointers) -----
    def evaluate_device(self, d_coeffs, n, z, stream=None):
        """polynomial_arithmetic::evaluate (polynomial_arithmetic.cpp:337-373)"""
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.lib.bbgpu_fr_evaluate_device(d_coeffs, n, _ptr(np.ascontiguousarray(z, dtype=np.uint64)), _ptr(out), stream or 0))
        return out

    def batch_invert_device(self, d_values, n, stream=None):
        """fr::batch_invert (fields/field.hpp:503-522), in place"""
        self._chk(self.lib.bbgpu_fr_batch_invert_device(d_values, n, stream or 0))

    def product_scan_device(self, d_in, d_out, n, reverse=False, inclusive=False, stream=None):
        """running products (prover.cpp:194-202 is the exclusive prefix form)"""
        self._chk(self.lib.bbgpu_fr_product_scan_device(d_in, d_out, n, int(reverse), int(inclusive), stream or 0))

    def mul_device(self, d_out, d_a, d_b, n, stream=None):
        """polynomial_arithmetic::mul (:328-335)"""
        self._chk(self.lib.bbgpu_fr_mul_device(d_out, d_a, d_b, n, stream or 0))

    def compute_kate_opening_coefficients_device(self, d_src, d_dest, n, z, stream=None):
        """polynomial_arithmetic::compute_kate_opening_coefficients (:562-591); returns F(z)"""
        f = np.zeros(4, dtype=np.uint64)
        self._chk(self.lib.bbgpu_kate_opening_device(d_src, d_dest, n, _ptr(np.ascontiguousarray(z, dtype=np.uint64)), _ptr(f), stream or 0))
        return f

    def compute_lagrange_polynomial_fft_device(self, d_l_1, n_src, n_target, stream=None):
        """polynomial_arithmetic::compute_lagrange_polynomial_fft (:381-476)"""
        self._chk(self.lib.bbgpu_lagrange_l1_fft_device(d_l_1, n_src, n_target, stream or 0))

    def divide_by_pseudo_vanishing_polynomial_device(self, d_coeffs, n_src, n_target, stream=None):
        """polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial (:478-560), in place"""
        self._chk(self.lib.bbgpu_divide_by_pseudo_vanishing_device(d_coeffs, n_src, n_target, stream or 0))

    def compute_permutation_lagrange_base_single_device(self, d_out, d_mapping, n, stream=None):
        """waffle::compute_permutation_lagrange_base_single (permutation.hpp:15-87)"""
        self._chk(self.lib.bbgpu_permutation_lagrange_base_device(d_out, d_mapping, n, stream or 0))

    # ---- scalar_multiplication ------------------------------------------------------------------------------------
    def srs_register(self, points_endo_table):
        return self._chk(self.lib.bbgpu_srs_register(_ptr(points_endo_table), points_endo_table.shape[0] // 2))

    def srs_generate(self, x_mont, n, want_host_table=False, first=0):
        """resident points x^(first + i) G, i < n (first > 0: the slice of a rank of a point-range split)"""
        table = np.zeros((2 * n, 8), dtype=np.uint64) if want_host_table else None
        xa = np.ascontiguousarray(x_mont, dtype=np.uint64)
        xp, tp = _ptr(xa), (_ptr(table) if want_host_table else None)
        h = self._chk(self.lib.bbgpu_srs_generate_range(xp, first, n, tp) if first else self.lib.bbgpu_srs_generate(xp, n, tp))
        return (h, table) if want_host_table else h

    def read_transcript(self, path, degree):
        """io::read_transcript (io.hpp:159-181), G1 part -> the (2 * degree, 8) endo table the reference's ReferenceString holds"""
        self.lib.bbgpu_transcript_read_g1.argtypes = [C.c_char_p, C.c_size_t, u64p]
        table = np.zeros((2 * degree, 8), dtype=np.uint64)
        self._chk(self.lib.bbgpu_transcript_read_g1(path.encode(), degree, _ptr(table)))
        return table

    def write_transcript(self, path, points_endo_table, degree, x_mont):
        """the file io::read_transcript accepts for this SRS: degree - 1 G1 points + {G2, x G2} (io.hpp:36-182)"""
        self.lib.bbgpu_transcript_write.argtypes = [C.c_char_p, u64p, C.c_size_t, u64p]
        self._chk(self.lib.bbgpu_transcript_write(path.encode(), _ptr(points_endo_table), degree, _ptr(np.ascontiguousarray(x_mont, dtype=np.uint64))))

    def srs_cache_stats(self):
        """(live tables, of which registered on first sight, device bytes those hold)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_uint64(0)
        self._chk(self.lib.bbgpu_srs_cache_stats(C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def memory_stats(self):
        """device / pinned bytes the library holds for the life of the process, as a dict (bbgpu_memory_stats)"""
        info = MemoryInfo()
        self._chk(self.lib.bbgpu_memory_stats(C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in MemoryInfo._fields_}

    def fault_inject(self, spec):
        """testing: "alloc:k" | "h2d:k" | "d2h:k" | "launch:k" makes the k-th such call from now on fail once; None disarms (bbgpu_fault_inject)"""
        self._chk(self.lib.bbgpu_fault_inject(spec.encode() if spec else None))

    def fault_stats(self):
        """funnel counters since the last spec, live device allocations, MSM slots in flight, as a dict (bbgpu_fault_stats)"""
        info = FaultInfo()
        self._chk(self.lib.bbgpu_fault_stats(C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in FaultInfo._fields_}

    def srs_set_validate(self, handle, full):
        """exact mode of the address-keyed table cache: host-pointer MSMs re-hash every row they use (handle -1: default for tables registered from now on)"""
        self._chk(self.lib.bbgpu_srs_set_validate(int(handle), 1 if full else 0))

    def srs_release(self, handle):
        self._chk(self.lib.bbgpu_srs_release(handle))

    def pippenger(self, scalars, points_endo_table, n=None):
        """returns the normalised g1::element (12,) -- or the infinity flag in y limb 3"""
        n = scalars.shape[0] if n is None else n
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.lib.bbgpu_msm_g1(_ptr(scalars), _ptr(points_endo_table), n, _ptr(out)))
        return out

    def pippenger_low_memory(self, scalars, points, n=None):
        """scalar_multiplication::pippenger_low_memory (:142-262): `points` is the PLAIN n-entry table (n x 8 u64), not the endo table"""
        n = scalars.shape[0] if n is None else n
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.lib.bbgpu_msm_g1_plain(_ptr(scalars), _ptr(points), n, _ptr(out)))
        return out

    def batched_scalar_multiplications(self, jobs):
        """jobs: list of (points_endo_table, scalars, n); returns list of normalised outputs"""
        arr = (MsmJob * len(jobs))()
        for j, (pts, sc, n) in zip(arr, jobs):
            j.points, j.scalars, j.num_elements = _ptr(pts), _ptr(sc), n
        self._chk(self.lib.bbgpu_msm_g1_batch(arr, len(jobs)))
        return [np.array(list(j.output), dtype=np.uint64) for j in arr]

    def msm_num_windows(self, n):
        return int(self.lib.bbgpu_msm_num_windows(n))

    def srs_num_windows(self, handle, n):
        return self._chk(self.lib.bbgpu_srs_num_windows(handle, n))

    def set_host_thresholds(self, msm_max_points, ntt_max_elements):
        """SURVEY 8b small sizes: host-pointer MSMs / transforms up to these sizes are answered on the host (0, 0: everything on the GPU)"""
        self.lib.bbgpu_set_host_thresholds(int(msm_max_points), int(ntt_max_elements))

    def set_table_share(self, rank, world):
        """tables registered from now on keep only the digit windows rank `rank` of `world` touches (multi-GPU row split); (0, 1): full tables"""
        self.lib.bbgpu_set_table_share(int(rank), int(world))

    def set_point_share(self, world):
        """tables registered from now on are a rank's n / world points of a larger MSM (point-range split): window size as for the whole; 1: default"""
        self.lib.bbgpu_set_point_share(int(world))

    def set_precompute(self, on=True):
        self.lib.bbgpu_set_precompute(1 if on else 0)

    def msm_device(self, handle, d_scalars_ptr, n, offset=0, window_begin=0, window_end=None, stream=None):
        if window_end is None:
            window_end = self.srs_num_windows(handle, n)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.lib.bbgpu_msm_g1_device(handle, offset, C.c_void_p(d_scalars_ptr), n, window_begin, window_end,
                                               _ptr(out), C.c_void_p(stream or 0)))
        return out

    def msm_device_async(self, handle, d_scalars_ptr, n, offset=0, window_begin=0, window_end=None, stream=None):
        """enqueue; returns a ticket for msm_wait().  At most eight MSMs in flight."""
        if window_end is None:
            window_end = self.srs_num_windows(handle, n)
        return self._chk(self.lib.bbgpu_msm_g1_device_async(handle, offset, C.c_void_p(d_scalars_ptr), n, window_begin, window_end,
                                                           C.c_void_p(stream or 0)))

    def msm_device_rows_async(self, handle, d_scalars_ptr, n, row_begin, row_end, offset=0, stream=None):
        """share [row_begin, row_end) of the W * n (window, point) pairs, window-major (row = w * n + i): a split that need not follow
        window boundaries (needs the window tables); returns a ticket for msm_wait()"""
        return self._chk(self.lib.bbgpu_msm_g1_device_rows_async(handle, offset, C.c_void_p(d_scalars_ptr), n, row_begin, row_end,
                                                                C.c_void_p(stream or 0)))

    def msm_device_buckets_async(self, handle, d_scalars_ptr, n, share, share_count, offset=0, stream=None):
        """share `share` of `share_count` of the BUCKET range (all windows, all points; needs complete window tables): 1 / N of the mixed
        additions and 1 / N of the bucket reduction; returns a ticket for msm_wait()"""
        self.lib.bbgpu_msm_g1_device_buckets_async.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        return self._chk(self.lib.bbgpu_msm_g1_device_buckets_async(handle, offset, C.c_void_p(d_scalars_ptr), n, share, share_count, C.c_void_p(stream or 0)))

    def srs_has_window_tables(self, handle):
        return self._chk(self.lib.bbgpu_srs_has_window_tables(handle)) == 1

    def msm_wait(self, ticket):
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.lib.bbgpu_msm_g1_wait(ticket, _ptr(out)))
        return out

    def msm_device_batch_async(self, handle, d_scalars_ptrs, n, offset=0, stream=None):
        """several MSMs over the same resident points as one pass (table mode); returns a ticket for msm_batch_wait()"""
        arr = (C.c_void_p * len(d_scalars_ptrs))(*d_scalars_ptrs)
        t = self._chk(self.lib.bbgpu_msm_g1_device_batch_async(handle, offset, arr, len(d_scalars_ptrs), n, C.c_void_p(stream or 0)))
        return t, len(d_scalars_ptrs)

    def msm_batch_wait(self, ticket):
        t, jobs = ticket
        out = np.zeros((jobs, 12), dtype=np.uint64)
        self._chk(self.lib.bbgpu_msm_g1_batch_wait(t, _ptr(out)))
        return out

    def g1_sum(self, points12):
        points12 = np.ascontiguousarray(points12, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.lib.bbgpu_g1_sum(_ptr(points12), points12.shape[0], _ptr(out)))
        return out

    # ---- host fallbacks of the drop-in boundary (what the C++ shim computes with after a GPU call has failed; no GPU needed) -----
    def host_msm(self, scalars, points, n=None, plain=False):
        n = scalars.shape[0] if n is None else n
        out = np.zeros(12, dtype=np.uint64)
        self.lib.bbgpu_host_msm_g1.argtypes = [u64p, u64p, C.c_size_t, C.c_int, u64p]
        self._chk(self.lib.bbgpu_host_msm_g1(_ptr(scalars), _ptr(points), n, int(plain), _ptr(out)))
        return out

    def host_ntt(self, coeffs, kind, constant=None):
        kind = NTT_KINDS[kind] if isinstance(kind, str) else kind
        cp = _ptr(np.ascontiguousarray(constant, dtype=np.uint64)) if constant is not None else None
        self.lib.bbgpu_host_ntt.argtypes = [u64p, C.c_size_t, C.c_int, u64p]
        self._chk(self.lib.bbgpu_host_ntt(_ptr(coeffs), coeffs.shape[0], kind, cp))
        return coeffs

    def host_evaluate(self, coeffs, z):
        out = np.zeros(4, dtype=np.uint64)
        self.lib.bbgpu_host_fr_evaluate.argtypes = [u64p, C.c_size_t, u64p, u64p]
        self._chk(self.lib.bbgpu_host_fr_evaluate(_ptr(coeffs), coeffs.shape[0], _ptr(np.ascontiguousarray(z, dtype=np.uint64)), _ptr(out)))
        return out

    def host_kate_opening(self, src, z):
        dest, f = np.zeros_like(src), np.zeros(4, dtype=np.uint64)
        self.lib.bbgpu_host_kate_opening.argtypes = [u64p, u64p, C.c_size_t, u64p, u64p]
        self._chk(self.lib.bbgpu_host_kate_opening(_ptr(src), _ptr(dest), src.shape[0], _ptr(np.ascontiguousarray(z, dtype=np.uint64)), _ptr(f)))
        return dest, f

    def host_lagrange_l1_fft(self, n_src, n_target):
        out = np.zeros((n_target, 4), dtype=np.uint64)
        self.lib.bbgpu_host_lagrange_l1_fft.argtypes = [u64p, C.c_size_t, C.c_size_t]
        self._chk(self.lib.bbgpu_host_lagrange_l1_fft(_ptr(out), n_src, n_target))
        return out

    def host_divide_by_pseudo_vanishing(self, coeffs, n_src, n_target):
        self.lib.bbgpu_host_divide_by_pseudo_vanishing.argtypes = [u64p, C.c_size_t, C.c_size_t]
        self._chk(self.lib.bbgpu_host_divide_by_pseudo_vanishing(_ptr(coeffs), n_src, n_target))
        return coeffs

    # ---- device self-test (known-answer entry points of the field / group layer) ---------------------------------------
    SELFTEST_FIELD_OPS = {"mul": 0, "sqr": 1, "add": 2, "sub": 3, "neg": 4, "mul_add": 5, "mul_sub": 6, "lazy_limbs": 7, "lazy_weak": 8,
                          "lazy_value": 9, "reduce": 10, "sqr_lazy": 11, "zero_tests": 12, "mul_addhi": 13, "sqr_addhi": 14}
    SELFTEST_G1_OPS = {"madd": 0, "add": 1, "dbl": 2, "dbl_affine": 3, "madd_neg": 4, "quad_add": 5}

    def selftest_field(self, field, op, a, b):
        """field: 'fq' | 'fr'; a, b: (n, 4) uint64 Montgomery operands -> (n, 4) canonical results of the DEVICE arithmetic"""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros_like(a)
        self.lib.bbgpu_selftest_field.argtypes = [C.c_int, C.c_int, u64p, u64p, C.c_size_t, u64p]
        self._chk(self.lib.bbgpu_selftest_field({"fq": 0, "fr": 1}[field], self.SELFTEST_FIELD_OPS[op], _ptr(a), _ptr(b), a.shape[0], _ptr(out)))
        return out

    def selftest_g1(self, op, p, q):
        """p, q: (n, 12) uint64 Jacobian -> (n, 16): X, Y, ZZ, ZZZ of the device result"""
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 12)
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 12)
        out = np.zeros((p.shape[0], 16), dtype=np.uint64)
        self.lib.bbgpu_selftest_g1.argtypes = [C.c_int, u64p, u64p, C.c_size_t, u64p]
        self._chk(self.lib.bbgpu_selftest_g1(self.SELFTEST_G1_OPS[op], _ptr(p), _ptr(q), p.shape[0], _ptr(out)))
        return out

    # ---- instrumentation ---------------------------------------------------------------------------------------------
    def set_timing(self, on=True):
        """False / 0: off; True / 1: an event after every stage; 2: only the accumulation kernel (light: for timed regions)"""
        self.lib.bbgpu_set_timing(int(on))

    def last_timing(self):
        buf = (C.c_float * 8)()
        k = self.lib.bbgpu_last_timing(buf, 8)
        return [float(buf[i]) for i in range(k)]
