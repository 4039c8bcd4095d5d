#!/bin/bash
# The batched forward of several workloads with list-filling forms 0 (the library's choice) and 4
# (cells), by kernel trace.
set -e
cd $GRAFT_REPO_ROOT
for WL in "$@"; do
  for FORM in 0 4; do
    bash tools/gpu_trace_form.sh $FORM --workload $WL | grep -v "^void raster"
  done
done
