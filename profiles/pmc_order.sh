set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_order_pmc
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r04_order_pmc/p1 -- python3 profiles/curvefit_order_probe.py > gpurun_out/r04_order_pmc/p1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r04_order_pmc/p1/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.OrderedDict()
for r in rows:
    if "curvefit_kernel" not in r["Kernel_Name"]: continue
    d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in d.items():
    print(k, "lane_util %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)), "valu_per_voxel %.0f" % (v["SQ_INSTS_VALU"] / 4194304), "valu_busy %.3f" % (v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]), "wait_any %.3f wait_inst %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
