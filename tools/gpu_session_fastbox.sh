#!/bin/bash
# The pool's boxes differ by up to 8 % on the render kernels.  This wrapper measures the split kernel first (a few seconds) and only
# runs the long artifact session (tools/gpu_session.sh: bench line + rocprofv3 stats + PMC, one consistent call) when the box is of
# the fast kind; otherwise it returns at once so that the call can be repeated on another box.
#   bash tools/gpu_session_fastbox.sh TAG MIN_KRAYS [session args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; MIN=$2; shift; shift
mkdir -p gpurun_out/$TAG
k=$(timeout 200 python tools/gpu_split_quickbench.py 2>/dev/null | head -1 | sed 's/.*-> *\([0-9.]*\) k rays.*/\1/')
echo "quickbench sn64: $k k rays/s (need >= $MIN)" | tee gpurun_out/$TAG/box.txt
if python -c "import sys; sys.exit(0 if float('${k:-0}') >= float('$MIN') else 1)"; then
    bash tools/gpu_session.sh $TAG "$@"
else
    echo "slow box: skipped"
fi
