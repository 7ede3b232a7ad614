#!/bin/bash
# gpurun wrapper: when no box is free (exit code 3: nothing ran, nothing charged) wait and ask again.
for i in 1 2 3 4 5 6 7 8; do
  gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
