"""Build libpnx_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m pyneapple_amd._build [--force] [--jobs N]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PNX_VARIANT=name builds a kernel-variant experiment beside the product library (libpnx_hip.name.so, own object dir);
# load it with PNX_LIB=<path> (pyneapple_amd/_lib.py)
_VARIANT = os.environ.get("PNX_VARIANT", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + _VARIANT if _VARIANT else ""))
LIB = os.path.join(HERE, f"libpnx_hip.{_VARIANT}.so" if _VARIANT else "libpnx_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = [*os.environ.get("PNX_EXTRA_FLAGS", "").split(), "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-ffp-contract=on", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]
N_MODELS = 7
# extra flags for the curve-fit translation units only (experiments: PNX_CURVEFIT_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp")
CURVEFIT_FLAGS = os.environ.get("PNX_CURVEFIT_FLAGS", "").split()
NNLS_FLAGS = os.environ.get("PNX_NNLS_FLAGS", "").split()  # e.g. -DPNX_NNLS_GBATCH=8 -DPNX_NNLS_WAVES_PER_SIMD=3 -DPNX_NNLS_LDS_ROWS=56


# kernel source groups: profiles/*_traffic.json / *_flops.json are stamped with source_id(group) of the build they were
# measured on, and bench.py only replays counters whose stamp matches the sources it is running
SOURCE_GROUPS = {
    "curvefit": ["pnx_curvefit_kernel.hpp", "pnx_curvefit_inst.hip"],
    "nnls": ["pnx_nnls.hip", "pnx_nnls.hpp", "pnx_nnls_dev.hpp", "pnx_nnls_qr.hip", "pnx_nnls_blk.hip", "pnx_nnls_blk_kernel.hpp"],
    "sweep": ["pnx_sweep.hip"],
    # the host boundary (streamed path, chunk ring, deferred NNLS hand-over, peak tables): host-mode / PCIe-inclusive figures and
    # the host-path fuzz summaries are stamped with this id, so that they are not replayed as current after pnx_api.hip changes
    # (round 4: the orchestration of both host paths moved into pnx_host_pipeline.hpp; the peak analysis the rings call is pnx_spectrum.hip)
    "host": ["pnx_api.hip", "pnx_internal.hpp", "pnx_host_pipeline.hpp", "pnx_spectrum.hip"],
}


def source_id(group: str) -> str:
    import hashlib

    h = hashlib.sha256()
    for f in SOURCE_GROUPS[group]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CXXFLAGS + (CURVEFIT_FLAGS if group == "curvefit" else NNLS_FLAGS if group == "nnls" else [])).encode())
    return h.hexdigest()[:16]


def source_ids() -> dict:
    return {g: source_id(g) for g in SOURCE_GROUPS}


def _units():
    units = [("pnx_api.o", "pnx_api.hip", []), ("pnx_nnls.o", "pnx_nnls.hip", NNLS_FLAGS), ("pnx_nnls_qr.o", "pnx_nnls_qr.hip", NNLS_FLAGS),
             ("pnx_nnls_blk.o", "pnx_nnls_blk.hip", NNLS_FLAGS),
             ("pnx_sweep.o", "pnx_sweep.hip", []), ("pnx_spectrum.o", "pnx_spectrum.hip", []),
             ("pnx_resize.o", "pnx_resize.hip", [])]
    for m in range(N_MODELS):
        units.append((f"pnx_curvefit_m{m}.o", "pnx_curvefit_inst.hip", [f"-DPNX_MODEL={m}", *CURVEFIT_FLAGS]))
    return [u for u in units if os.path.exists(os.path.join(CSRC, u[1]))]


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    d.append(os.path.join(HERE, "..", "include", "pnx.h"))
    d.append(os.path.abspath(__file__))
    return d


def _compile(unit):
    obj, src, extra = unit
    cmd = [HIPCC, *CXXFLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", os.path.join(OBJ, obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode == 0:
        with open(os.path.join(OBJ, obj + ".cmd"), "w") as fh:
            fh.write(" ".join(cmd))
    return obj, r.returncode, (r.stdout + r.stderr)


def _stale(unit, force: bool) -> bool:
    """An object is rebuilt when it is missing, older than its source or any header, or was compiled
    with another command line (so a one-kernel edit recompiles one translation unit, not the seven curve-fit ones)."""
    obj, src, extra = unit
    o = os.path.join(OBJ, obj)
    if force or not os.path.exists(o):
        return True
    cmd = " ".join([HIPCC, *CXXFLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", o])
    try:
        with open(o + ".cmd") as fh:
            if fh.read() != cmd:
                return True
    except OSError:
        return True
    deps = [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "pnx.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    if src.startswith("pnx_curvefit") or src in ("pnx_sweep.hip", "pnx_spectrum.hip", "pnx_resize.hip"):
        deps = [d for d in deps if "pnx_nnls" not in os.path.basename(d)]  # these units include no NNLS header
    return max(os.path.getmtime(d) for d in deps) > os.path.getmtime(o)


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    units = _units()
    objs = [os.path.join(OBJ, u[0]) for u in units]
    if _VARIANT:
        # a variant differs from the product in the flags of a few units: link the product's objects for the others
        base = os.path.join(CSRC, "_obj")
        keep = []
        for i, (obj, src, extra) in enumerate(units):
            po = os.path.join(base, obj)
            cmd = " ".join([HIPCC, *CXXFLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", po])
            try:
                same = open(po + ".cmd").read() == cmd
            except OSError:
                same = False
            if same and not force:
                objs[i] = po
            else:
                keep.append((obj, src, extra))
        units = keep
    todo = [u for u in units if _stale(u, force)]
    if not todo and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(o) for o in objs):
        return LIB
    jobs = jobs or min(max(1, len(todo)), max(1, (os.cpu_count() or 2)))
    with cf.ThreadPoolExecutor(jobs) as ex:
        for obj, rc, out in ex.map(_compile, todo):
            if verbose or rc:
                sys.stderr.write(f"[pnx build] {obj}: rc={rc}\n{out}\n")
            if rc:
                raise RuntimeError(f"hipcc failed for {obj}")
    # the dynamic symbol table is the header's function list and nothing else: a linker version script made from include/pnx.h
    # (-fvisibility=hidden alone leaves the C++ runtime's template instantiations and the kernels' host-side handles visible)
    import re

    with open(os.path.join(HERE, "..", "include", "pnx.h")) as fh:
        api = sorted(set(re.findall(r"^PNX_API\s+int\s+(pnx_[a-z0-9_]+)\s*\(", fh.read(), flags=re.M)))
    vscript = os.path.join(OBJ, "pnx_exports.map")
    with open(vscript, "w") as fh:
        fh.write("{\n  global:\n" + "".join(f"    {n};\n" for n in api) + "  local:\n    *;\n};\n")
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={vscript}", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("link of libpnx_hip.so failed")
    return LIB


if __name__ == "__main__":
    j = None
    if "--jobs" in sys.argv:
        j = int(sys.argv[sys.argv.index("--jobs") + 1])
    print(build(force="--force" in sys.argv, jobs=j, verbose=True))
