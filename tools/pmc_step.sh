#!/bin/bash
# Hardware counters of every kernel of one training step, three rocprofv3 passes (counters are not multiplexed):
#   a: matrix-core / wave / LDS activity (8 SQ counters)   b: bytes fetched from HBM   c: bytes written + L2 hits
#   tools/pmc_step.sh <tag> <bench.py arguments...>      results: gpurun_out/pmc_step/<tag>_{a,b,c}
#   (pass --also "" --also_modes "" to profile the headline configuration only)
# Summarise with tools/pmc_step_summary.py gpurun_out/pmc_step/<tag>.
set -e
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_step
mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_a -- python3 bench.py "$@" > $out/${tag}_a.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_b -- python3 bench.py "$@" > $out/${tag}_b.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/${tag}_c -- python3 bench.py "$@" > $out/${tag}_c.log 2>&1
echo "done $tag"
