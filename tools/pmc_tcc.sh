#!/bin/bash
# L2 (TCC) hit / miss counters for one convolution shape; one rocprofv3 pass.
#   tools/pmc_tcc.sh TAG H W Cin Cout k stride pad batch fwd|dgrad|wgrad
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag="$1"; shift
out=gpurun_out/pmc
mkdir -p $out
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $out/${tag}_tcc -- python3 tools/pmc_conv.py "$@" > $out/${tag}_tcc.log 2>&1
echo "done $tag"
