"""The wave-resident LDE kernel's index model (tools/lde_wave_model.py): both transforms against a plain NTT, the four LDS address
functions against the ds_write_b64 / ds_read_b64 banking rules, the store pattern of the forward transform.  The kernel
(csrc/kernels_lde.hip: lde_columns_wave_kernel) builds its tables and addresses by the same formulas; the GPU tests compare its
output with the oracle bit for bit (tests/test_gpu_kernels.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import lde_wave_model as M  # noqa: E402


def test_model_transforms_and_bank_rules():
    M.run_checks(seed=1, rate_bits=2)


def test_model_with_another_rate_and_seed():
    M.run_checks(seed=7, rate_bits=1)


def test_radix2_twiddle_exponents_are_the_kernels_constants():
    """lde_w32_exp in the kernel: 2^(78 k) forward, 2^(114 k) inverse (w_64 = 2^39)."""
    for inverse, base in ((False, 78), (True, 114)):
        assert [M.pow2_exponent(v) for v in M.tables(inverse)[3]] == [(base * k) % 192 for k in range(16)]
