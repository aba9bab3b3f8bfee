"""ThreadSanitizer and AddressSanitizer / UBSan over the host-side orchestration of PNX_MEM_HOST calls
(pyneapple_amd/csrc/pnx_host_pipeline.hpp: the chunk ring's IN / LAUNCH / OUT / page-touch stages and the state machine around
the one streamed kernel of a host-array curve fit).  The header is free of HIP types; tests/host_stub/host_pipeline_stub.cpp
instantiates it on a stub device (streams = worker threads, the persistent kernel = a thread polling the watermark and the
abort word) with failure injection in every stage and a stalled upload.  Sanitizers on the CPU build only (the pool offers no
GPU sanitizers); the product's HIP instantiation of the same code is what the `-m gpu` suite runs."""
from __future__ import annotations

import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "host_stub", "host_pipeline_stub.cpp")


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_orchestration_is_clean_under_sanitizers(tmp_path, sanitizer):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / ("stub_" + sanitizer.replace(",", "_")))
    cmd = ["g++", "-std=c++17", "-g", "-O1", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-pthread",
           "-I" + os.path.join(ROOT, "pyneapple_amd", "csrc"), SRC, "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode and any(s in b.stderr for s in ("cannot find -ltsan", "cannot find -lasan", "cannot find -lubsan")):
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-4000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0 and "host pipeline stub ok" in r.stdout, out[-4000:]
