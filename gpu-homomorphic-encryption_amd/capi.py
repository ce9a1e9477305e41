"""ctypes binding of include/fhe_hip.h.  Thin by design: every call goes straight to the C ABI and
raises FheError on a non-zero status (no CPU fallback anywhere)."""
import ctypes
import os

import numpy as np

from .build import library_path

WIDTH_32, WIDTH_64, WIDTH_52, WIDTH_256, WIDTH_64X = 1, 2, 3, 4, 5
_lib = None

U64x4 = ctypes.c_uint64 * 4


class FheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fhe_hip error {code}: {msg}")
        self.code = code


def lib():
    """Load libfhe_hip.so.  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise FheError(-100, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the HIP engine)")
    L = ctypes.CDLL(path)
    vp, u32, u64, sz, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    P = ctypes.POINTER
    sigs = {
        "fhe_hip_abi_version": ([], ci),
        "fhe_hip_last_error": ([], ctypes.c_char_p),
        "fhe_hip_device_count": ([P(ci)], ci),
        "fhe_hip_set_device": ([ci], ci),
        "fhe_hip_get_device": ([P(ci)], ci),
        "fhe_hip_device_name": ([ctypes.c_char_p, sz], ci),
        "fhe_hip_malloc": ([P(vp), sz], ci),
        "fhe_hip_free": ([vp], ci),
        "fhe_hip_memset": ([vp, ci, sz], ci),
        "fhe_hip_memcpy_h2d": ([vp, vp, sz], ci),
        "fhe_hip_memcpy_d2h": ([vp, vp, sz], ci),
        "fhe_hip_memcpy_d2d": ([vp, vp, sz], ci),
        "fhe_hip_sync": ([], ci),
        "fhe_montgomery_inverse": ([U64x4, U64x4], ci),
        "fhe_montgomery_params": ([U64x4, U64x4, U64x4], ci),
        "fhe_find_ntt_primes": ([u32, u32, u32, P(u64)], ci),
        "fhe_find_ntt_primes_wide": ([u32, u32, u32, vp], ci),
        "fhe_find_psi": ([u32, U64x4, U64x4], ci),
        "fhe_u256_add_mod": ([vp, vp, vp, U64x4, sz, vp], ci),
        "fhe_u256_sub_mod": ([vp, vp, vp, U64x4, sz, vp], ci),
        "fhe_u256_mont_mul": ([vp, vp, vp, U64x4, u64, sz, vp], ci),
        "fhe_u256_mont_mul_scalar": ([vp, vp, U64x4, U64x4, u64, sz, vp], ci),
        "fhe_ntt_create": ([P(vp), u32, U64x4], ci),
        "fhe_ntt_destroy": ([vp], ci),
        "fhe_ntt_set_stream": ([vp, vp], ci),
        "fhe_ntt_width_class": ([vp], ci),
        "fhe_ntt_forward": ([vp, vp, u32], ci),
        "fhe_ntt_inverse": ([vp, vp, u32], ci),
        "fhe_ntt_pointwise": ([vp, vp, vp, vp, u32], ci),
        "fhe_ntt_multiply": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_ntt_create": ([P(vp), u32, vp, u32], ci),
        "fhe_rns_ntt_destroy": ([vp], ci),
        "fhe_rns_ntt_set_stream": ([vp, vp], ci),
        "fhe_rns_ntt_width_class": ([vp], ci),
        "fhe_rns_ntt_reserve": ([vp, u32], ci),
        "fhe_rns_ntt_workspace_bytes": ([vp, ctypes.POINTER(ctypes.c_uint64)], ci),
        "fhe_rns_ntt_forward": ([vp, vp, u32], ci),
        "fhe_rns_ntt_inverse": ([vp, vp, u32], ci),
        "fhe_rns_ntt_pointwise": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_ntt_multiply": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_poly_add": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_poly_sub": ([vp, vp, vp, vp, u32], ci),
        "fhe_ct_multiply": ([vp] * 8 + [u32], ci),
        "fhe_rns_check_canonical": ([vp, vp, u32], ci),
        "fhe_rns_to_rns": ([vp, vp, vp, u32], ci),
        "fhe_rns_from_rns": ([vp, vp, vp, u32], ci),
        "fhe_rns_monomial_mul_sub": ([vp, vp, vp, vp, u32], ci),
        "fhe_blind_rotate_step": ([vp, vp, vp, vp, vp, vp, vp, vp, u32], ci),
        "fhe_blind_rotate": ([vp, P(vp), P(vp), u32, vp, vp, vp, vp, vp, u32], ci),
        "fhe_rns_fast_base_convert": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_base_create": ([P(vp), vp, u32], ci),
        "fhe_rns_ntt_multiply_bcast": ([vp, vp, vp, vp, u32], ci),
        "fhe_rns_mul_mont_literal": ([vp, vp, vp, vp, u32], ci),
        "fhe_ref_forward_kernel_literal": ([vp, vp, U64x4, u64, u32, u32, vp], ci),
        "fhe_ref_inverse_kernel_literal": ([vp, vp, U64x4, u64, U64x4, u32, u32, vp], ci),
        "fhe_ref_stockham_stage_literal": ([vp, vp, vp, U64x4, u64, u32, u32, u32, vp], ci),
        "fhe_bit_reverse": ([vp, u32, u32, vp], ci),
        "fhe_sample_uniform_lcg": ([vp, U64x4, u64, sz, vp], ci),
        "fhe_sample_gaussian_placeholder": ([vp, U64x4, u64, sz, vp], ci),
        "fhe_rns_sample_ternary": ([vp, vp, ctypes.c_double, u64, u32], ci),
        "fhe_rns_sample_gaussian": ([vp, vp, ctypes.c_double, u64, u32], ci),
        "fhe_rns_sample_uniform": ([vp, vp, u64, u32], ci),
        "fhe_gaussian_cdt": ([ctypes.c_double, P(u64), u32, P(u32)], ci),
        "fhe_poly_mod_switch": ([vp, vp, U64x4, U64x4, sz, vp], ci),
        "fhe_negacyclic_reduce": ([vp, U64x4, sz, vp], ci),
        "fhe_rns_rescale_drop_last": ([vp, vp, vp, u32], ci),
        "fhe_relin_num_digits": ([vp, u32, P(u32)], ci),
        "fhe_relin_keys_create": ([vp, P(vp), u32, P(vp), P(vp), u32], ci),
        "fhe_relin_keys_destroy": ([vp], ci),
        "fhe_ct_relinearize": ([vp, vp, vp, vp, vp, u32], ci),
        "fhe_ct_multiply_relin": ([vp, vp, vp, vp, vp, vp, vp, vp, u32], ci),
        "fhe_timer_create": ([P(vp)], ci),
        "fhe_timer_destroy": ([vp], ci),
        "fhe_rns_timer_start": ([vp, vp], ci),
        "fhe_rns_timer_stop": ([vp, vp], ci),
        "fhe_timer_elapsed_ms": ([vp, P(ctypes.c_float)], ci),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(L, name)          # AttributeError here == the library does not export a declared symbol
        fn.argtypes, fn.restype = args, res
    L._fhe_symbols = sorted(sigs)
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise FheError(rc, lib().fhe_hip_last_error().decode(errors="replace"))


def _q4(q):
    return U64x4(*[(int(q) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])


def _int(v):
    return sum(int(v[i]) << (64 * i) for i in range(4))


# ---- host-side parameter maths (works without a GPU) -------------------------------------------
def montgomery_inverse(q):
    out = U64x4(); _check(lib().fhe_montgomery_inverse(_q4(q), out)); return _int(out)


def montgomery_params(q):
    r2, inv = U64x4(), U64x4(); _check(lib().fhe_montgomery_params(_q4(q), r2, inv)); return _int(r2), _int(inv)


def find_ntt_primes(bits, n, count):
    if bits > 64:
        out = (U64x4 * count)(); _check(lib().fhe_find_ntt_primes_wide(bits, n, count, ctypes.cast(out, ctypes.c_void_p)))
        return [_int(x) for x in out]
    out = (ctypes.c_uint64 * count)(); _check(lib().fhe_find_ntt_primes(bits, n, count, out)); return [int(x) for x in out]


def find_psi(n, q):
    out = U64x4(); _check(lib().fhe_find_psi(n, _q4(q), out)); return _int(out)


def device_count():
    c = ctypes.c_int(0)
    rc = lib().fhe_hip_device_count(ctypes.byref(c))
    return c.value if rc == 0 else 0


def device_name():
    buf = ctypes.create_string_buffer(256); _check(lib().fhe_hip_device_name(buf, 256)); return buf.value.decode()


def sync():
    _check(lib().fhe_hip_sync())


# ---- device memory --------------------------------------------------------------------------------
class DeviceBuffer:
    """hipMalloc'd buffer of 256-bit containers; host view = numpy uint64 (..., 4)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _check(lib().fhe_hip_malloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes); b.upload(arr); return b

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(lib().fhe_hip_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, shape=None, dtype=np.uint64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().fhe_hip_sync())      # engines enqueue asynchronously; callers synchronise (tests/test_fhe.cu:88)
        _check(lib().fhe_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out.reshape(shape) if shape is not None else out.reshape(-1, 4)

    def zero(self):
        _check(lib().fhe_hip_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if getattr(self, "ptr", None):
            lib().fhe_hip_free(self.ptr); self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):           # torch tensor on the GPU
        return x.data_ptr()
    return int(x)


# ---- literal element-wise primitives ----------------------------------------------------------------
def ref_forward_kernel_literal(data, twiddles, q, inv0, n, batch=1, stream=None):
    _check(lib().fhe_ref_forward_kernel_literal(_ptr(data), _ptr(twiddles), _q4(q), inv0, n, batch, stream))


def ref_inverse_kernel_literal(data, inv_twiddles, q, inv0, n_inv, n, batch=1, stream=None):
    _check(lib().fhe_ref_inverse_kernel_literal(_ptr(data), _ptr(inv_twiddles), _q4(q), inv0, _q4(n_inv), n, batch, stream))


def ref_stockham_stage_literal(output, data, twiddles, q, inv0, n, stage, batch=1, stream=None):
    _check(lib().fhe_ref_stockham_stage_literal(_ptr(output), _ptr(data), _ptr(twiddles), _q4(q), inv0, n, stage, batch, stream))


def bit_reverse(data, n, batch=1, stream=None):
    _check(lib().fhe_bit_reverse(_ptr(data), n, batch, stream))


def sample_uniform_lcg(out, q, seed, count, stream=None):
    _check(lib().fhe_sample_uniform_lcg(_ptr(out), _q4(q), seed, count, stream))


def sample_gaussian_placeholder(out, q, seed, count, stream=None):
    _check(lib().fhe_sample_gaussian_placeholder(_ptr(out), _q4(q), seed, count, stream))


def gaussian_cdt(sigma):
    n = ctypes.c_uint32(0); _check(lib().fhe_gaussian_cdt(sigma, None, 0, ctypes.byref(n)))
    t = (ctypes.c_uint64 * n.value)(); _check(lib().fhe_gaussian_cdt(sigma, t, n.value, ctypes.byref(n)))
    return [int(v) for v in t]


def poly_mod_switch(r, a, old_q, new_q, count, stream=None):
    _check(lib().fhe_poly_mod_switch(_ptr(r), _ptr(a), _q4(old_q), _q4(new_q), count, stream))


def negacyclic_reduce(data, q, n, stream=None):
    _check(lib().fhe_negacyclic_reduce(_ptr(data), _q4(q), n, stream))


def u256_add_mod(r, a, b, q, count, stream=None):
    _check(lib().fhe_u256_add_mod(_ptr(r), _ptr(a), _ptr(b), _q4(q), count, stream))


def u256_sub_mod(r, a, b, q, count, stream=None):
    _check(lib().fhe_u256_sub_mod(_ptr(r), _ptr(a), _ptr(b), _q4(q), count, stream))


def u256_mont_mul(r, a, b, q, inv0, count, stream=None):
    _check(lib().fhe_u256_mont_mul(_ptr(r), _ptr(a), _ptr(b), _q4(q), inv0, count, stream))


def u256_mont_mul_scalar(r, a, scalar, q, inv0, count, stream=None):
    _check(lib().fhe_u256_mont_mul_scalar(_ptr(r), _ptr(a), _q4(scalar), _q4(q), inv0, count, stream))


# ---- engines ------------------------------------------------------------------------------------------
class NttEngine:
    """fhe::NTTEngine (include/ntt.cuh:72-103)."""

    def __init__(self, n, q):
        self.h = None
        h = ctypes.c_void_p()
        _check(lib().fhe_ntt_create(ctypes.byref(h), n, _q4(q)))
        self.h, self.n, self.q = h, n, q

    @property
    def width_class(self):
        return lib().fhe_ntt_width_class(self.h)

    def set_stream(self, stream):
        _check(lib().fhe_ntt_set_stream(self.h, stream))

    def forward(self, d_data, batch=1):
        _check(lib().fhe_ntt_forward(self.h, _ptr(d_data), batch))

    def inverse(self, d_data, batch=1):
        _check(lib().fhe_ntt_inverse(self.h, _ptr(d_data), batch))

    def pointwise(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_ntt_pointwise(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def multiply(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_ntt_multiply(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def close(self):
        if self.h:
            lib().fhe_ntt_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RnsNttEngine:
    """fhe::RNS_NTTEngine (include/ntt.cuh:106-137) + the FHEContext::multiply tensor product."""

    def __init__(self, n, moduli):
        self.h = None
        L = len(moduli)
        arr = (U64x4 * L)(*[_q4(q) for q in moduli])
        h = ctypes.c_void_p()
        if n is None:                      # RNS base without a ring (RNSContext): degree-1 engine, buffers [count][L]
            _check(lib().fhe_rns_base_create(ctypes.byref(h), ctypes.cast(arr, ctypes.c_void_p), L)); n = 1
        else:
            _check(lib().fhe_rns_ntt_create(ctypes.byref(h), n, ctypes.cast(arr, ctypes.c_void_p), L))
        self.h, self.n, self.L, self.moduli = h, n, L, list(moduli)

    @property
    def width_class(self):
        return lib().fhe_rns_ntt_width_class(self.h)

    def set_stream(self, stream):
        _check(lib().fhe_rns_ntt_set_stream(self.h, stream))

    def forward(self, d_data, batch=1):
        _check(lib().fhe_rns_ntt_forward(self.h, _ptr(d_data), batch))

    def inverse(self, d_data, batch=1):
        _check(lib().fhe_rns_ntt_inverse(self.h, _ptr(d_data), batch))

    def pointwise(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_rns_ntt_pointwise(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def multiply(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_rns_ntt_multiply(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def multiply_bcast(self, d_r, d_a, d_b_one, batch=1):
        _check(lib().fhe_rns_ntt_multiply_bcast(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b_one), batch))

    def poly_add(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_rns_poly_add(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def mul_mont_literal(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_rns_mul_mont_literal(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def poly_sub(self, d_r, d_a, d_b, batch=1):
        _check(lib().fhe_rns_poly_sub(self.h, _ptr(d_r), _ptr(d_a), _ptr(d_b), batch))

    def ct_multiply(self, d_c0, d_c1, d_c2, d_a0, d_a1, d_b0, d_b1, batch=1):
        _check(lib().fhe_ct_multiply(self.h, _ptr(d_c0), _ptr(d_c1), _ptr(d_c2), _ptr(d_a0), _ptr(d_a1), _ptr(d_b0),
                                     _ptr(d_b1), batch))

    def to_rns(self, d_rns, d_values, batch=1):
        _check(lib().fhe_rns_to_rns(self.h, _ptr(d_rns), _ptr(d_values), batch))

    def from_rns(self, d_values, d_rns, batch=1):
        _check(lib().fhe_rns_from_rns(self.h, _ptr(d_values), _ptr(d_rns), batch))

    def monomial_mul_sub(self, d_out, d_in, d_shifts, batch=1):
        _check(lib().fhe_rns_monomial_mul_sub(self.h, _ptr(d_out), _ptr(d_in), _ptr(d_shifts), batch))

    def blind_rotate_step(self, rows_c0, rows_c1, d_acc0, d_acc1, d_shifts, d_tmp0, d_tmp1, batch=1):
        _check(lib().fhe_blind_rotate_step(self.h, rows_c0.h, rows_c1.h, _ptr(d_acc0), _ptr(d_acc1), _ptr(d_shifts), _ptr(d_tmp0),
                                           _ptr(d_tmp1), batch))

    def blind_rotate(self, rows_c0, rows_c1, d_acc0, d_acc1, d_shifts, d_tmp0, d_tmp1, batch=1):
        """rows_c0 / rows_c1: one imported row set per step (lists of equal length); d_shifts: device uint32 [steps][batch]."""
        steps = len(rows_c0); assert len(rows_c1) == steps
        arr = ctypes.c_void_p * max(steps, 1)
        a0 = arr(*[r.h for r in rows_c0]); a1 = arr(*[r.h for r in rows_c1])
        _check(lib().fhe_blind_rotate(self.h, a0, a1, steps, _ptr(d_acc0), _ptr(d_acc1), _ptr(d_shifts), _ptr(d_tmp0), _ptr(d_tmp1), batch))

    def sample_ternary(self, d_out, probability, seed, batch=1):
        _check(lib().fhe_rns_sample_ternary(self.h, _ptr(d_out), probability, seed, batch))

    def sample_gaussian(self, d_out, sigma, seed, batch=1):
        _check(lib().fhe_rns_sample_gaussian(self.h, _ptr(d_out), sigma, seed, batch))

    def sample_uniform(self, d_out, seed, batch=1):
        _check(lib().fhe_rns_sample_uniform(self.h, _ptr(d_out), seed, batch))

    def fast_base_convert(self, target, d_out, d_in, batch=1):
        _check(lib().fhe_rns_fast_base_convert(self.h, target.h, _ptr(d_out), _ptr(d_in), batch))

    def rescale_drop_last(self, d_out, d_in, batch=1):
        _check(lib().fhe_rns_rescale_drop_last(self.h, _ptr(d_out), _ptr(d_in), batch))

    def relin_num_digits(self, decomp_bits):
        k = ctypes.c_uint32(0); _check(lib().fhe_relin_num_digits(self.h, decomp_bits, ctypes.byref(k))); return k.value

    def import_relin_keys(self, decomp_bits, keys_b, keys_a):
        """keys_b / keys_a: lists of device buffers, each an [L][n] polynomial in coefficient form."""
        n = len(keys_b)
        pb = (ctypes.c_void_p * n)(*[_ptr(k) for k in keys_b]); pa = (ctypes.c_void_p * n)(*[_ptr(k) for k in keys_a])
        out = ctypes.c_void_p()
        _check(lib().fhe_relin_keys_create(self.h, ctypes.byref(out), decomp_bits, pb, pa, n))
        return RelinKeys(out)

    def relinearize(self, rk, d_c0, d_c1, d_c2, batch=1):
        _check(lib().fhe_ct_relinearize(self.h, rk.h, _ptr(d_c0), _ptr(d_c1), _ptr(d_c2), batch))

    def reserve(self, batch):
        """Pre-size the library workspaces for calls of up to `batch` units (needed before hipGraph capture)."""
        _check(lib().fhe_rns_ntt_reserve(self.h, batch))

    def workspace_bytes(self):
        """Device bytes the engine's library-owned workspaces hold right now."""
        out = ctypes.c_uint64(0)
        _check(lib().fhe_rns_ntt_workspace_bytes(self.h, ctypes.byref(out)))
        return out.value

    def ct_multiply_relin(self, rk, d_c0, d_c1, d_a0, d_a1, d_b0, d_b1, batch=1):
        """FHEContext::multiply: tensor product + relinearisation in one call (two components out)."""
        _check(lib().fhe_ct_multiply_relin(self.h, rk.h, _ptr(d_c0), _ptr(d_c1), _ptr(d_a0), _ptr(d_a1), _ptr(d_b0), _ptr(d_b1), batch))

    def check_canonical(self, d_data, batch=1):
        _check(lib().fhe_rns_check_canonical(self.h, _ptr(d_data), batch))

    def close(self):
        if self.h:
            lib().fhe_rns_ntt_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RelinKeys:
    """Relinearisation keys imported into an engine (NTT-domain copies owned by the library)."""

    def __init__(self, h):
        self.h = h

    def close(self):
        if self.h:
            lib().fhe_relin_keys_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Timer:
    """hipEvent pair recorded on an RnsNttEngine's stream."""

    def __init__(self):
        self.t = None
        t = ctypes.c_void_p(); _check(lib().fhe_timer_create(ctypes.byref(t))); self.t = t

    def start(self, eng):
        _check(lib().fhe_rns_timer_start(eng.h, self.t))

    def stop(self, eng):
        _check(lib().fhe_rns_timer_stop(eng.h, self.t))

    def elapsed_ms(self):
        ms = ctypes.c_float(0); _check(lib().fhe_timer_elapsed_ms(self.t, ctypes.byref(ms))); return ms.value

    def __del__(self):
        try:
            if self.t:
                lib().fhe_timer_destroy(self.t); self.t = None
        except Exception:
            pass
