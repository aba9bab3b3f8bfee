#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in an object file or library built by pyneapple_amd/_build.py:
    python tools/kernel_resources.py pyneapple_amd/csrc/_obj/pnx_nnls_blk.o [name-filter]
Reads the AMDGPU metadata note of the embedded gfx950 code object (no GPU needed)."""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    out = []
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "dev.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={path}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            # a host object: the fat binary sits in section .hip_fatbin
            fb = os.path.join(d, "fat.bin")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout)
    return out


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    for notes in code_objects(path):
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            if flt and flt not in name:
                continue
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} vgpr_spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} "
                  f"sgpr_spill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
