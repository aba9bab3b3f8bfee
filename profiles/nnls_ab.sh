set -e
out=gpurun_out/$1.txt; shift
for v in "$@"; do
  lib=""; [ "$v" != "product" ] && lib=pyneapple_amd/libpnx_hip.$v.so
  timeout -k 10 150 python profiles/nnls_probe.py $lib >> $out 2>&1
  tail -1 $out
done
