#!/bin/bash
# Local launcher: 1 ps + N workers on this machine (default N=2), the working
# counterpart of the reference's recognition_flower/distributed/run.sh (which
# starts two copies of worker 0 and no ps).
#   ./run.sh [num_workers] [extra train_dataset.py flags...]
set -e
cd "$(dirname "$0")"
# peer-mapped GPU memory (--ps_backend=xgmi, RCCL) needs the dmabuf IPC path of the ROCm runtime
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NW=${1:-2}
shift || true
PS_PORT=${PS_PORT:-2223}
WORKER_PORT0=${WORKER_PORT0:-2230}
workers=""
for ((i = 0; i < NW; i++)); do
  workers+="\"localhost:$((WORKER_PORT0 + i))\""
  [ $i -lt $((NW - 1)) ] && workers+=","
done
cluster="{\"ps\":[\"localhost:${PS_PORT}\"],\"worker\":[${workers}]}"
pids=()
TF_CONFIG="{\"cluster\":${cluster},\"task\":{\"index\":0,\"type\":\"ps\"}}" python train_dataset.py "$@" &
pids+=($!)
for ((i = 0; i < NW; i++)); do
  TF_CONFIG="{\"cluster\":${cluster},\"task\":{\"index\":${i},\"type\":\"worker\"}}" python train_dataset.py "$@" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do
  wait "$p" || rc=$?
done
exit $rc
