#!/bin/bash
for cap in 8 16 32 64; do
echo "cap $cap"
VFD_KSPLIT_CAP=$cap python tools/layer_bench.py --only "enc.final" 2>/dev/null | grep -v "^layer\|totals"
VFD_KSPLIT_CAP=$cap python tools/layer_bench.py --only "D.cls" 2>/dev/null | grep -v "^layer\|totals"
VFD_KSPLIT_CAP=$cap python tools/layer_bench.py --only "dec.init" 2>/dev/null | grep -v "^layer\|totals"
done
