#!/bin/bash
# NOTE: stream_block_rows exists in the sources of commit 214bccb only (removed: below the review's bar); the comparison
# library was PNX_VARIANT=nostream PNX_NNLS_FLAGS=-DPNX_BLK_STREAM_ROWS=0 python -m pyneapple_amd._build of that commit.
# Round 5: the four-slot block kernel with block rows 8 .. 15 streamed through the block sweeps (product) against the same
# source with -DPNX_BLK_STREAM_ROWS=0 (rows >= 64 row by row, as up to round 4): strong regularisers, 2^18 voxels each.
set -e
out=gpurun_out/r05_stream_ab.txt
for lib in "" pyneapple_amd/libpnx_hip.nostream.so; do
  echo "== ${lib:-product}" >> $out
  PNX_LIB=${lib:+$PWD/$lib} timeout -k 10 400 python profiles/nnls_mu_probe.py 1,0.5 1,0.2 2,0.5 3,0.1 2,0.1 2>&1 | grep "block plan" | sed 's/"checksum.*//' >> $out
done
cat $out
