#!/bin/bash
# One gpurun call: runs the commands given as arguments (each a quoted shell line) from the repo root on the GPU box and keeps
# their output under gpurun_out/<tag>_<n>.log.  Usage: gpurun -- 'bash tools/gpu_run.sh <tag> "<cmd>" "<cmd>" ...'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
tag="$1"; shift
n=0
for cmd in "$@"; do
  n=$((n+1))
  echo "== [$tag $n] $cmd"
  ( eval "$cmd" ) > "gpurun_out/${tag}_${n}.log" 2>&1
  echo "rc=$? ($(wc -l < gpurun_out/${tag}_${n}.log) lines)"
  grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" "gpurun_out/${tag}_${n}.log" | tail -n 25
done
