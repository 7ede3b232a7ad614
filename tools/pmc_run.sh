#!/bin/bash
# Two counter passes (8 SQ counters each) over one convolution shape.
#   tools/pmc_run.sh H W Cin Cout k stride pad batch fwd|dgrad|wgrad
# Each pass is its own rocprofv3 run (counters are not multiplexed); results land in gpurun_out/pmc/<tag>_{a,b}.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag="$1_$2_$3_$4_k$5_s$6_b$8_$9"
out=gpurun_out/pmc
mkdir -p $out
timeout -s KILL 60 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/${tag}_a -- python3 tools/pmc_conv.py "$@" > $out/${tag}_a.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $out/${tag}_b -- python3 tools/pmc_conv.py "$@" > $out/${tag}_b.log 2>&1
echo "done $tag"
