#!/bin/bash
# One gpurun call: stamps the commit the snapshot was taken from (.commit: the box has no .git), then runs the given command
# there.  usage: bash scripts/gpu.sh <timeout seconds> '<command>'
t=$1; shift
echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- hydrium_amd bench.py scripts || echo +dirty)" > .commit
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
