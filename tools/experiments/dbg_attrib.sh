#!/bin/bash
# Main-loop attribution of the 3x3 layers: times tools/bench_conv.py rows under the timing-only -DDTFA_DBG=n builds
# (tools/build_variant.py dbg<n> -DDTFA_DBG=<n> beforehand).  usage: dbg_attrib.sh <batch> <rows> [extra bench_conv flags]
set -e
B=${1:-2048}; ROWS=${2:-9,14,19}; shift 2 || true
mkdir -p gpurun_out
echo "== full" 
timeout -k 10 120 python tools/bench_conv.py --batch $B --only $ROWS --iters 10 "$@"
for n in 1 2 4 8 9 13 15; do
  echo "== DTFA_DBG=$n"
  DTFA_HIP_LIB=variants/dbg$n/libdtfa_hip.so timeout -k 10 120 python tools/bench_conv.py --batch $B --only $ROWS --iters 10 "$@"
done
