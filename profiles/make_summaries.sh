#!/bin/bash
# Turn the raw rocprofv3 output of one profiling round (gpurun_out/<x>/{kt,traffic,pmc_cf,pmc_nnls}, bench.json) into the
# committed summaries profiles/<tag>_*.   usage: bash profiles/make_summaries.sh gpurun_out/h r01_h
set -e
src=$1; tag=$2
python3 - "$src" "$tag" <<'PY'
import csv, glob, sys
src, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{src}/kt/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as o:
    w = csv.writer(o); w.writerow(rows[0])
    for r in rows[1:]:
        if "pnx::" not in r[0]:
            r[0] = r[0][:60] + "..."
        w.writerow(r)
for r in rows[1:]:
    if "pnx::" in r[0]:
        print(r[0][:80], r[1], r[3])
PY
python3 profiles/traffic_summary.py $src/traffic > profiles/${tag}_traffic.json
(echo "# curvefit_kernel<4,5,true,false,false,false> (MODEL, N, FD, PV, T1, STREAM), full C3 volume, 1 timed + 1 warm-up launch (profiles/pmc_curvefit.sh)"; python3 profiles/pmc_summary.py $src/pmc_cf curvefit_kernel) > profiles/${tag}_pmc_curvefit.txt
(echo "# nnls_blk_kernel (pnx_nnls_blk.hip), full C4 volume = TWO dispatches per step since r04_r (the pilot of 12 288 voxels and the launch over the other 4 182 016: per-dispatch figures are the mean of the two, i.e. per 2 097 152 voxels; four launches of 2^20 before), profiles/pmc_nnls.sh"; python3 profiles/pmc_summary.py $src/pmc_nnls "nnls_blk_kernel") > profiles/${tag}_pmc_nnls.txt
(echo "# nnls_aty_mfma_kernel (profiles/pmc_traffic.sh, mfma pass)"; python3 profiles/pmc_summary.py $src/traffic/mfma nnls_aty) > profiles/${tag}_pmc_mfma_aty.txt
cp $src/bench.json profiles/${tag}_bench.json
python3 - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
def parse(fn):
    d = {}
    for line in open(fn):
        m = re.match(r"(\w+)\s+dispatches=\s*(\d+) sum=([\d.e+]+) per_dispatch=([\d.e+]+)", line)
        if m: d[m.group(1)] = float(m.group(4))
    return d
out = {}
for name, fn, nvox in (("curvefit_kernel<4, 5, true, false, false, false>", f"profiles/{tag}_pmc_curvefit.txt", 4194304), ("nnls_blk_kernel", f"profiles/{tag}_pmc_nnls.txt", 4194304 // 2)):
    d = parse(fn)
    f64 = d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_ADD_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d.get("SQ_INSTS_VALU_TRANS_F64", 0)
    fl = 64 * (f64 + d["SQ_INSTS_VALU_FMA_F64"])
    out[name] = {"voxels_per_launch": nvox, "fp64_flop_per_launch_issued": fl, "fp64_flop_per_voxel_issued": fl / nvox,
                 "lane_utilisation": d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64),
                 "fp64_share_of_valu_instructions": f64 / d["SQ_INSTS_VALU"], "valu_instructions_per_voxel": d["SQ_INSTS_VALU"] / nvox,
                 "salu_instructions_per_voxel": d.get("SQ_INSTS_SALU", 0) / nvox,
                 "valu_issue_busy": d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], "source": fn,
                 "note": "64 lanes counted for every issued fp64 VALU instruction (FMA = 2 flop); multiply by lane_utilisation for flops on active lanes"}
import os, sys
sys.path.insert(0, os.getcwd())
from pyneapple_amd import _build
ids = _build.source_ids()  # bench.py replays these counters only while the kernel sources still hash to this
out["_source_ids"] = ids
json.dump(out, open(f"profiles/{tag}_flops.json", "w"), indent=1)
t = json.load(open(f"profiles/{tag}_traffic.json"))
t["_source_ids"] = ids
t["_nnls_launch_voxels"] = 4194304 // 2
json.dump(t, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict) and not k.startswith("_"):
        print(k[:30], {a: round(b, 3) for a, b in v.items() if isinstance(b, float)})
b = json.load(open(f"profiles/{tag}_bench.json"))
print("bench:", b["value"], b["secondary"]["value"], b["roofline_sweep"]["frac"], b["roofline_mfma"]["frac"])
PY
