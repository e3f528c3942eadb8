#!/bin/bash
LIB=aeonflux_amd/lib/libaeonflux_gpu.so
cp $LIB /tmp/ab_default.so
for r in 1 2 3; do
  for which in /tmp/ab_default.so variants/pinned_first.so; do
    cp $which $LIB
    echo "== $which"
    python tools/mixed_concurrency.py 64 16 2>&1 | head -1 | cut -c1-160
    python tools/mixed_concurrency.py 64 1 2>&1 | head -1 | cut -c1-160
  done
done
cp /tmp/ab_default.so $LIB
