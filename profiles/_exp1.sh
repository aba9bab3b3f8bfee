export PNX_NNLS_GENERAL=1
for v in g0 ka kb kc kd ke; do timeout -k 10 200 python profiles/nnls_probe.py pyneapple_amd/libpnx_hip.$v.so | cut -c1-100; done
