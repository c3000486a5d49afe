#!/bin/bash
# gpurun wrapper: records the commit the snapshot is taken at (.git_head: the GPU box has no .git; the profile tools
# stamp it into the files they write) and forwards its arguments:   tools/gpu.sh --timeout 900 -- 'bash tools/profile_round.sh r05'
cd "$(dirname "$0")/.." && (git rev-parse --short=12 HEAD; git status --porcelain | grep -q . && echo "+dirty") | tr -d '\n' > .git_head
exec /usr/local/graft/bin/gpurun "$@"
