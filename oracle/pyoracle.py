"""ctypes view of the CPU oracle (oracle/fhe_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does (see oracle/fhe_oracle.h for the parity-pinning statement).

Arrays of 256-bit values are numpy uint64 arrays of shape (..., 4), little-endian limbs --
byte-identical to the reference's `uint256_t` arrays (include/bigint.cuh:9-11).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FHE_ORACLE_ASAN=1 loads the AddressSanitizer / UBSan build (`make -C oracle asan`; run python with LD_PRELOAD=$(gcc -print-file-name=libasan.so))
_LIB_PATH = os.path.join(_HERE, "_build", "libfhe_oracle_asan.so" if os.environ.get("FHE_ORACLE_ASAN") == "1" else "libfhe_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (no GPU, no reference files needed)."""
    src = [os.path.join(_HERE, f) for f in ("fhe_oracle.c", "fhe_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


class U256(ctypes.Structure):
    _fields_ = [("limbs", ctypes.c_uint64 * 4)]


def _p(arr):
    return arr.ctypes.data_as(ctypes.POINTER(U256))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        P = ctypes.POINTER(U256)
        u64, u32, sz, ci, vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p
        sig = {
            "orc_add_mod": (None, [P, P, P, P]),
            "orc_sub_mod": (None, [P, P, P, P]),
            "orc_mont_mul": (None, [P, P, P, P, u64]),
            "orc_mont_inverse": (u64, [P]),
            "orc_ct_butterfly": (None, [P, P, P, P, u64]),
            "orc_gs_butterfly": (None, [P, P, P, P, u64]),
            "orc_batch_add": (None, [P, P, P, P, sz]),
            "orc_batch_sub": (None, [P, P, P, P, sz]),
            "orc_batch_mont": (None, [P, P, P, P, u64, sz]),
            "orc_ref_forward_kernel": (None, [P, P, P, u64, u32]),
            "orc_ref_inverse_kernel": (None, [P, P, P, u64, P, u32]),
            "orc_ref_pointwise_kernel": (None, [P, P, P, P, u64, u32]),
            "orc_ref_placeholder_table": (None, [P, u32]),
            "orc_ref_stockham_stage": (None, [P, P, P, P, u64, u32, u32]),
            "orc_plan_create": (vp, [u32, P]),
            "orc_plan_destroy": (None, [vp]),
            "orc_plan_psi": (None, [vp, P]),
            "orc_plan_inv0": (u64, [vp]),
            "orc_plan_twiddle": (None, [vp, u32, P]),
            "orc_ntt_forward": (None, [vp, P]),
            "orc_ntt_inverse": (None, [vp, P]),
            "orc_ntt_pointwise": (None, [vp, P, P, P]),
            "orc_polymul_ntt": (None, [vp, P, P, P]),
            "orc_polymul_schoolbook": (None, [vp, P, P, P]),
            "orc_rns_forward": (ci, [ctypes.POINTER(vp), u32, P, u32, ci]),
            "orc_rns_inverse": (ci, [ctypes.POINTER(vp), u32, P, u32, ci]),
            "orc_rns_polymul": (ci, [ctypes.POINTER(vp), u32, P, P, P, u32, ci]),
            "orc_ct_multiply": (ci, [ctypes.POINTER(vp), u32] + [P] * 7 + [u32, ci]),
            "orc_max_threads": (ci, []),
            "orc_rns_polymul_narrow": (ci, [ctypes.POINTER(vp), u32, P, P, P, u32, ci]),
            "orc_sample_uniform_lcg": (None, [P, P, u64, sz]),
            "orc_sample_gaussian_placeholder": (None, [P, P, u64, sz]),
            "orc_ctr_rand": (u64, [u64, u64, u64]),
            "orc_sample_ternary": (None, [ctypes.POINTER(vp), u32, P, u64, u64, u32]),
            "orc_gaussian_cdt": (u32, [ctypes.c_double, ctypes.POINTER(u64), u32]),
            "orc_sample_gaussian": (None, [ctypes.POINTER(vp), u32, P, ctypes.POINTER(u64), u32, u64, u32]),
            "orc_sample_uniform": (None, [ctypes.POINTER(vp), u32, P, u64, u32]),
            "orc_poly_mod_switch": (None, [P, P, P, u64, sz]),
            "orc_negacyclic_reduce": (None, [P, P, sz]),
            "orc_to_rns": (None, [ctypes.POINTER(vp), u32, P, P, u32]),
            "orc_from_rns": (ci, [ctypes.POINTER(vp), u32, P, P, u32]),
            "orc_monomial_mul_sub": (None, [ctypes.POINTER(vp), u32, P, P, ctypes.POINTER(u32), u32]),
            "orc_fast_base_convert": (None, [ctypes.POINTER(vp), u32, ctypes.POINTER(vp), u32, P, P, u32]),
            "orc_rescale_drop_last": (None, [ctypes.POINTER(vp), u32, P, P, u32]),
            "orc_relin_num_digits": (u32, [ctypes.POINTER(vp), u32, u32]),
            "orc_relinearize": (ci, [ctypes.POINTER(vp), u32, u32, P, P, P, ctypes.POINTER(P), ctypes.POINTER(P), u32, ci]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


# ---- conversions ---------------------------------------------------------------------------
MASK64 = (1 << 64) - 1


def to_limbs(values):
    """Python ints -> (len, 4) uint64 array (values taken mod 2^256)."""
    vals = list(values)
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (v >> (64 * k)) & MASK64
    return out


def from_limbs(arr):
    """(..., 4) uint64 array -> flat list of Python ints."""
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def small_to_limbs(a64):
    """uint64 array of residues < 2^64 -> (..., 4) container array with zero upper limbs."""
    a64 = np.asarray(a64, dtype=np.uint64)
    out = np.zeros(a64.shape + (4,), dtype=np.uint64)
    out[..., 0] = a64
    return out


def _one(v):
    return to_limbs([v])


# ---- scalar primitives on Python ints --------------------------------------------------------
def add_mod(a, b, q):
    r = np.zeros((1, 4), np.uint64); A, B, Q = _one(a), _one(b), _one(q)
    lib().orc_add_mod(_p(r), _p(A), _p(B), _p(Q)); return from_limbs(r)[0]


def sub_mod(a, b, q):
    r = np.zeros((1, 4), np.uint64); A, B, Q = _one(a), _one(b), _one(q)
    lib().orc_sub_mod(_p(r), _p(A), _p(B), _p(Q)); return from_limbs(r)[0]


def mont_inverse(q):
    Q = _one(q); return int(lib().orc_mont_inverse(_p(Q)))


def mont_mul(a, b, q, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    r = np.zeros((1, 4), np.uint64); A, B, Q = _one(a), _one(b), _one(q)
    lib().orc_mont_mul(_p(r), _p(A), _p(B), _p(Q), inv0); return from_limbs(r)[0]


def ct_butterfly(a, b, w, q, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    A, B, W, Q = _one(a), _one(b), _one(w), _one(q)
    lib().orc_ct_butterfly(_p(A), _p(B), _p(W), _p(Q), inv0); return from_limbs(A)[0], from_limbs(B)[0]


def gs_butterfly(a, b, w, q, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    A, B, W, Q = _one(a), _one(b), _one(w), _one(q)
    lib().orc_gs_butterfly(_p(A), _p(B), _p(W), _p(Q), inv0); return from_limbs(A)[0], from_limbs(B)[0]


# ---- batch primitives on limb arrays ---------------------------------------------------------
def _chk(*arrs):
    for a in arrs:
        assert a.dtype == np.uint64 and a.flags.c_contiguous and a.shape[-1] == 4


def batch_add(a, b, q):
    _chk(a, b); r = np.empty_like(a); Q = _one(q)
    lib().orc_batch_add(_p(r), _p(a), _p(b), _p(Q), a.size // 4); return r


def batch_sub(a, b, q):
    _chk(a, b); r = np.empty_like(a); Q = _one(q)
    lib().orc_batch_sub(_p(r), _p(a), _p(b), _p(Q), a.size // 4); return r


def batch_mont(a, b, q, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    _chk(a, b); r = np.empty_like(a); Q = _one(q)
    lib().orc_batch_mont(_p(r), _p(a), _p(b), _p(Q), inv0, a.size // 4); return r


# ---- literal reference kernels ---------------------------------------------------------------
def ref_placeholder_table(n):
    t = np.zeros((n, 4), np.uint64); lib().orc_ref_placeholder_table(_p(t), n); return t


def ref_forward_kernel(data, tw, q, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    _chk(data, tw); d = data.copy(); Q = _one(q)
    lib().orc_ref_forward_kernel(_p(d), _p(tw), _p(Q), inv0, d.shape[0]); return d


def ref_inverse_kernel(data, itw, q, n_inv, inv0=None):
    if inv0 is None:
        inv0 = mont_inverse(q)
    _chk(data, itw); d = data.copy(); Q, NI = _one(q), _one(n_inv)
    lib().orc_ref_inverse_kernel(_p(d), _p(itw), _p(Q), inv0, _p(NI), d.shape[0]); return d


def ref_stockham_stage(data, tw, q, stage, inv0=None):
    """ntt_stockham_kernel (kernels/ntt_kernels.cu:213-243): one out-of-place stage over one polynomial."""
    if inv0 is None:
        inv0 = mont_inverse(q)
    _chk(data, tw); out = data.copy(); Q = _one(q)
    lib().orc_ref_stockham_stage(_p(out), _p(data), _p(tw), _p(Q), inv0, data.shape[0], stage); return out


def ref_pointwise_kernel(a, b, q, inv0=None):
    return batch_mont(a, b, q, inv0)


# ---- intended-maths engine ---------------------------------------------------------------------
class Plan:
    """One (n, q) negacyclic NTT plan == what NTTEngine's constructor should have computed."""

    def __init__(self, n, q):
        Q = _one(q)
        self.h = lib().orc_plan_create(n, _p(Q))
        if not self.h:
            raise ValueError(f"oracle: no plan for n={n}, q={q} (need odd prime q = 1 mod 2n, q < 2^255)")
        self.n, self.q = n, q

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_plan_destroy(self.h); self.h = None

    @property
    def psi(self):
        r = np.zeros((1, 4), np.uint64); lib().orc_plan_psi(self.h, _p(r)); return from_limbs(r)[0]

    @property
    def inv0(self):
        return int(lib().orc_plan_inv0(self.h))

    def twiddle(self, k):
        r = np.zeros((1, 4), np.uint64); lib().orc_plan_twiddle(self.h, k, _p(r)); return from_limbs(r)[0]

    def forward(self, data):
        _chk(data); d = data.copy(); assert d.shape[0] == self.n
        lib().orc_ntt_forward(self.h, _p(d)); return d

    def inverse(self, data):
        _chk(data); d = data.copy(); assert d.shape[0] == self.n
        lib().orc_ntt_inverse(self.h, _p(d)); return d

    def pointwise(self, a, b):
        _chk(a, b); r = np.empty_like(a); lib().orc_ntt_pointwise(self.h, _p(r), _p(a), _p(b)); return r

    def polymul(self, a, b):
        _chk(a, b); r = np.empty_like(a); lib().orc_polymul_ntt(self.h, _p(r), _p(a), _p(b)); return r

    def schoolbook(self, a, b):
        _chk(a, b); r = np.empty_like(a); lib().orc_polymul_schoolbook(self.h, _p(r), _p(a), _p(b)); return r


class RnsPlan:
    """RNS_NTTEngine analogue: data limb-major [batch][L][n] (src/ntt.cu:161)."""

    def __init__(self, n, moduli):
        self.plans = [Plan(n, q) for q in moduli]
        self.n, self.L, self.moduli = n, len(moduli), list(moduli)
        self._arr = (ctypes.c_void_p * self.L)(*[p.h for p in self.plans])

    def _batch(self, a):
        _chk(a); assert a.size % (4 * self.L * self.n) == 0
        return a.size // (4 * self.L * self.n)

    def forward(self, data, threads=1):
        d = data.copy(); lib().orc_rns_forward(self._arr, self.L, _p(d), self._batch(d), threads); return d

    def inverse(self, data, threads=1):
        d = data.copy(); lib().orc_rns_inverse(self._arr, self.L, _p(d), self._batch(d), threads); return d

    def polymul(self, a, b, threads=1):
        r = np.empty_like(a)
        self.threads_used = lib().orc_rns_polymul(self._arr, self.L, _p(r), _p(a), _p(b), self._batch(a), threads)
        return r

    def ct_multiply(self, a0, a1, b0, b1, threads=1):
        c0, c1, c2 = np.empty_like(a0), np.empty_like(a0), np.empty_like(a0)
        lib().orc_ct_multiply(self._arr, self.L, _p(c0), _p(c1), _p(c2), _p(a0), _p(a1), _p(b0), _p(b1),
                              self._batch(a0), threads)
        return c0, c1, c2


def _rns_polymul_narrow(self, a, b, threads=1):
    """Word-sized CPU port (q < 2^62) of polymul: same outputs, 64-bit arithmetic."""
    _chk(a, b); out = np.empty_like(a)
    used = lib().orc_rns_polymul_narrow(self._arr, self.L, _p(out), _p(a), _p(b), self._batch(a), threads)
    if used < 0:
        raise ValueError("polymul_narrow needs moduli below 2^62")
    self.threads_used = used
    return out


def _rns_relin(self, decomp_bits, c0, c1, c2, keys_b, keys_a, threads=1):
    """c0', c1' (copies) after key switching c2 with the L*K keys (each a [L][n] limb array, coefficient form)."""
    P = ctypes.POINTER(U256)
    K = int(lib().orc_relin_num_digits(self._arr, self.L, decomp_bits))
    assert len(keys_b) == len(keys_a) == self.L * K, (len(keys_b), self.L, K)
    kb = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys_b]; ka = [np.ascontiguousarray(k, dtype=np.uint64) for k in keys_a]
    pb = (P * len(kb))(*[_p(k) for k in kb]); pa = (P * len(ka))(*[_p(k) for k in ka])
    o0, o1 = c0.copy(), c1.copy()
    rc = lib().orc_relinearize(self._arr, self.L, decomp_bits, _p(o0), _p(o1), _p(np.ascontiguousarray(c2)), pb, pa, self._batch(c2), threads)
    assert rc > 0
    return o0, o1


def _rns_num_digits(self, decomp_bits):
    return int(lib().orc_relin_num_digits(self._arr, self.L, decomp_bits))


def _rns_to_rns(self, values):
    """[batch][n] 256-bit integers -> [batch][L][n] residues."""
    _chk(values); batch = values.size // (4 * self.n)
    out = np.empty((batch, self.L, self.n, 4), np.uint64)
    lib().orc_to_rns(self._arr, self.L, _p(out), _p(np.ascontiguousarray(values)), batch); return out


def _rns_from_rns(self, rns):
    """[batch][L][n] residues -> [batch][n] integers in [0, Q) (CRT)."""
    batch = self._batch(rns)
    out = np.empty((batch, self.n, 4), np.uint64)
    rc = lib().orc_from_rns(self._arr, self.L, _p(out), _p(np.ascontiguousarray(rns)), batch)
    if rc != 0:
        raise ValueError("oracle: product of the moduli must stay below 2^255")
    return out


def _rns_rescale(self, rns):
    """[batch][L][n] -> [batch][L-1][n]: round(c / q_last), limb-wise."""
    batch = self._batch(rns); assert self.L >= 2
    out = np.empty((batch, self.L - 1, self.n, 4), np.uint64)
    lib().orc_rescale_drop_last(self._arr, self.L, _p(out), _p(np.ascontiguousarray(rns)), batch); return out


def _rns_base_convert(self, target, rns):
    """Fast base conversion of [batch][L][n] residues to the basis of `target` (another RnsPlan): [batch][L'][n]."""
    batch = self._batch(rns); assert target.n == self.n
    out = np.empty((batch, target.L, self.n, 4), np.uint64)
    lib().orc_fast_base_convert(self._arr, self.L, target._arr, target.L, _p(out), _p(np.ascontiguousarray(rns)), batch); return out


def _rns_monomial_mul_sub(self, rns, shifts):
    """(X^shift[b] - 1) * p for every polynomial of ciphertext b."""
    batch = self._batch(rns); sh = np.ascontiguousarray(shifts, dtype=np.uint32); assert sh.size == batch
    out = np.empty_like(rns)
    lib().orc_monomial_mul_sub(self._arr, self.L, _p(out), _p(np.ascontiguousarray(rns)), sh.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), batch)
    return out


def _rns_blind_rotate_step(self, decomp_bits, acc0, acc1, shifts, rows0, rows1, threads=1):
    """acc + ExternalProduct((X^a - 1) * acc, RGSW): rows0 / rows1 = (keys_b, keys_a) for the digits of component 0 / 1."""
    d0 = self.monomial_mul_sub(acc0, shifts); d1 = self.monomial_mul_sub(acc1, shifts)
    o0, o1 = self.relinearize(decomp_bits, acc0, acc1, d0, rows0[0], rows0[1], threads)
    return self.relinearize(decomp_bits, o0, o1, d1, rows1[0], rows1[1], threads)


def _rns_blind_rotate(self, decomp_bits, acc0, acc1, shifts, rows0_list, rows1_list, threads=1):
    """The loop: shifts is [steps][batch]; rows*_list hold one RGSW row set per step."""
    for sh, r0, r1 in zip(np.asarray(shifts, dtype=np.uint32), rows0_list, rows1_list):
        acc0, acc1 = self.blind_rotate_step(decomp_bits, acc0, acc1, sh, r0, r1, threads)
    return acc0, acc1


RnsPlan.polymul_narrow = _rns_polymul_narrow
RnsPlan.monomial_mul_sub = _rns_monomial_mul_sub
RnsPlan.blind_rotate = _rns_blind_rotate
RnsPlan.blind_rotate_step = _rns_blind_rotate_step
RnsPlan.fast_base_convert = _rns_base_convert
RnsPlan.rescale_drop_last = _rns_rescale
RnsPlan.to_rns = _rns_to_rns
RnsPlan.from_rns = _rns_from_rns
RnsPlan.relinearize = _rns_relin
RnsPlan.num_digits = _rns_num_digits


# ---- row N4: samplers, single-modulus modulus switch, negacyclic fold ------------------------------------------
def sample_uniform_lcg(q, seed, count):
    """sample_uniform_kernel, literal (src/polynomial.cu:130-143)."""
    out = np.empty((count, 4), dtype=np.uint64); lib().orc_sample_uniform_lcg(_p(out), _p(_one(q)), seed, count); return out


def sample_gaussian_placeholder(q, seed, count):
    """sample_gaussian_kernel, literal placeholder (src/polynomial.cu:113-128)."""
    out = np.empty((count, 4), dtype=np.uint64); lib().orc_sample_gaussian_placeholder(_p(out), _p(_one(q)), seed, count); return out


def ctr_rand(seed, index, draw):
    return int(lib().orc_ctr_rand(seed, index, draw))


def gaussian_cdt(sigma):
    n = lib().orc_gaussian_cdt(sigma, None, 0)
    t = (ctypes.c_uint64 * n)(); lib().orc_gaussian_cdt(sigma, t, n)
    return [int(v) for v in t]


def poly_mod_switch(a, old_q, new_q):
    _chk(a); out = np.empty_like(a); lib().orc_poly_mod_switch(_p(out), _p(a), _p(_one(old_q)), new_q, a.size // 4); return out


def negacyclic_reduce(data, q):
    _chk(data); d = data.copy(); lib().orc_negacyclic_reduce(_p(d), _p(_one(q)), d.size // 8); return d


def _rns_sample_ternary(self, probability, seed, batch=1):
    out = np.empty((batch, self.L, self.n, 4), dtype=np.uint64)
    lib().orc_sample_ternary(self._arr, self.L, _p(out), int(probability * 4294967296.0), seed, batch); return out


def _rns_sample_gaussian(self, sigma, seed, batch=1):
    t = gaussian_cdt(sigma); arr = (ctypes.c_uint64 * len(t))(*t)
    out = np.empty((batch, self.L, self.n, 4), dtype=np.uint64)
    lib().orc_sample_gaussian(self._arr, self.L, _p(out), arr, len(t), seed, batch); return out


def _rns_sample_uniform(self, seed, batch=1):
    out = np.empty((batch, self.L, self.n, 4), dtype=np.uint64)
    lib().orc_sample_uniform(self._arr, self.L, _p(out), seed, batch); return out


RnsPlan.sample_ternary = _rns_sample_ternary
RnsPlan.sample_gaussian = _rns_sample_gaussian
RnsPlan.sample_uniform = _rns_sample_uniform


def max_threads():
    return int(lib().orc_max_threads())
