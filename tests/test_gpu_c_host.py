"""The C ABI from a host with no Python and no torch in the process: examples/host.cpp is built with hipcc (only
for hipMalloc / hipMemcpy) and run against the in-tree libnesr_hip.so."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_runs_the_forward(tmp_path, cuda_device):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "nesr_host")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "host.cpp"), "-o", exe, "-ldl"], check=True, timeout=300)
    lib = os.path.join(ROOT, "neural_enhanced_super_resolution_amd", "libnesr_hip.so")
    out = subprocess.run([exe, lib], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "max difference 0 LSB, non-finite 0" in out.stdout
