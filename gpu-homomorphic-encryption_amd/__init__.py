"""MI355X-native RNS-NTT polynomial-multiply engine (gfx950), behind the call surface of
codebasecomprehension987/gpu-homomorphic-encryption's fhe::NTTEngine / RNS_NTTEngine /
PolynomialOps / FHEContext::multiply.

The product is the C-ABI shared library ``lib/libfhe_hip.so`` (headers: ``include/fhe_hip.h`` and the
C++ mirror ``include/fhe/*.hpp``).  This Python package is only the loader / ctypes view used by the
tests and by bench.py; it never falls back to a CPU implementation.

The directory name carries a hyphen, so import it by string:
    importlib.import_module("gpu-homomorphic-encryption_amd")
"""
from .build import build_library, library_path  # noqa: F401
from .capi import (  # noqa: F401
    FheError, NttEngine, RnsNttEngine, DeviceBuffer, Timer, lib, device_count, find_ntt_primes, find_psi,
    montgomery_inverse, montgomery_params, u256_add_mod, u256_sub_mod, u256_mont_mul, u256_mont_mul_scalar,
    ref_forward_kernel_literal, ref_inverse_kernel_literal, ref_stockham_stage_literal, bit_reverse, sample_uniform_lcg, sample_gaussian_placeholder, gaussian_cdt, poly_mod_switch, negacyclic_reduce,
    WIDTH_32, WIDTH_52, WIDTH_64, WIDTH_256, WIDTH_64X,
)

__all__ = [
    "build_library", "library_path", "FheError", "NttEngine", "RnsNttEngine", "DeviceBuffer", "Timer", "lib",
    "device_count", "find_ntt_primes", "find_psi", "montgomery_inverse", "montgomery_params", "u256_add_mod",
    "u256_sub_mod", "u256_mont_mul", "u256_mont_mul_scalar", "ref_forward_kernel_literal", "ref_inverse_kernel_literal", "ref_stockham_stage_literal", "bit_reverse", "sample_uniform_lcg", "sample_gaussian_placeholder",
    "gaussian_cdt", "poly_mod_switch", "negacyclic_reduce", "WIDTH_32", "WIDTH_52", "WIDTH_64", "WIDTH_256", "WIDTH_64X",
]
