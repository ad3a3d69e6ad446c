import os
import sys

import pytest

# One ROCm runtime per process: some tests use torch (device memory for the reference-shaped launches, gloo).
# torch bundles its own libamdhip64 / librccl; if the product library were loaded first it would bring in
# /opt/rocm's copies and `import torch` would then load a SECOND HIP runtime.  Importing torch first makes the
# product library (NEEDED libamdhip64.so.7, dlopen librccl.so.1) bind to the copies torch already loaded.
try:
    import torch  # noqa: F401
except ImportError:  # the product itself does not need torch
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, like __graft_entry__.build()
    lib = os.path.join(ROOT, "ppa-nbody-collisions_amd", "libnbody_mi355x.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ppa-nbody-collisions_amd", "csrc"), "all"])
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libnbody_ref.so (built where /root/reference exists)")


def _gpu_present():
    # device files only: importing torch / touching HIP here would initialise the GPU in the test runner
    return os.path.exists("/dev/kfd") and os.path.isdir("/dev/dri")


def pytest_collection_modifyitems(config, items):
    import oracle_lib
    skip_ref = pytest.mark.skip(reason="oracle/_ref/libnbody_ref.so not built (no /root/reference here)")
    skip_gpu = pytest.mark.skip(reason="no GPU device files on this host")
    for item in items:
        if "ref" in item.keywords and not oracle_lib.have_ref():
            item.add_marker(skip_ref)
        if "gpu" in item.keywords and not _gpu_present():
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def nb():
    """The product package (ctypes over libnbody_mi355x.so). Import fails loudly if the library is absent."""
    import ppa_nbody_collisions_amd as m
    return m
