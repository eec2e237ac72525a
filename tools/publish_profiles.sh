#!/bin/bash
# gpurun_out/<tag>/summary.txt (tools/collect.sh) -> profiles/<name>_rocprof_summary.txt (tracked) + profiles/traffic.json[key]
#   tools/publish_profiles.sh <tag> <name> <traffic key> <steps> [note] [kernel filter]
set -eu
cd "$(dirname "$0")/.."
T=$1; N=$2; K=$3; S=$4; NOTE=${5:-}; F=${6:-}
grep -v "at::native" gpurun_out/$T/summary.txt | sed "s#gpurun_out/$T#(gpurun_out/$T on the GPU box)#" > profiles/${N}_rocprof_summary.txt
grep -h "fwd+bwd\|\"metric\"" gpurun_out/$T/stats.log 2>/dev/null | cut -c1-400 | sed 's/^/# stdout of the profiled command: /' >> profiles/${N}_rocprof_summary.txt || true
python3 tools/update_traffic.py "$K" profiles/${N}_rocprof_summary.txt "$S" "$NOTE" "$F"
