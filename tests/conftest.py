"""pytest configuration.

Markers
  gpu : needs a real MI355X (run by the driver with `-m gpu` through gpurun); everything else must
        pass on a CPU-only box.

Fixtures
  oracle  : the CPU checker (oracle/bn254_ref.py) -- test infrastructure only
  emu     : tests/emu/libg16_emu.so = the kernel sources compiled against the SIMT emulator
            (test infrastructure only; never loaded by the package)
  gpulib  : the product library libg16_amd.so (skips if it is not built)
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The suites pin the sort / bucket path for every Prover that does not ask otherwise: since round 5 the
# library's own default sends small single-device keys through fixed-base tables (g16_options.fixed_tables
# = 0, "automatic"), which would leave the bucket kernels without their small-size cases.  The table
# path has its own tests (tables=1 / tables=0 explicitly); bench.py and smoke() run the product default.
import circom_compat_amd as _cc  # noqa: E402

_cc.DEFAULT_TABLES = -1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run via gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on the CPU")


@pytest.fixture(scope="session")
def oracle():
    import bn254_ref
    return bn254_ref


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def emu():
    """Kernel sources built against the CPU SIMT emulator (tests only)."""
    # G16_EMU_LIB: another build of the same emulator library (scripts/asan_emu.sh: the kernels under
    # AddressSanitizer -- device memory is host malloc there, so an out-of-bounds kernel access is a report)
    if os.environ.get("G16_EMU_LIB"):
        from circom_compat_amd import _binding
        return _binding.Library(os.environ["G16_EMU_LIB"])
    so = os.path.join(ROOT, "tests", "emu", "libg16_emu.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "circom_compat_amd", "csrc"), "emu", "-j8"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("emulation build failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    from circom_compat_amd import _binding
    return _binding.Library(so)


@pytest.fixture(scope="session")
def gpulib():
    """The product library.  On a GPU box a missing build is a hard failure (no silent fallback)."""
    from circom_compat_amd import _binding
    return _binding.load()


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue("emu" if request.param == "emu" else "gpulib")
