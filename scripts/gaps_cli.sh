#!/bin/bash
# idle gaps of the GPU during the shipped loop (rocprofv3 kernel trace of scripts/cli_run.py, rocpd database), by the kernel that follows the gap
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps_cli
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/gaps_cli -o p -- python $GRAFT_REPO_ROOT/scripts/cli_run.py ${1:-100} 2>/dev/null | tail -1
F=$(find /tmp/gaps_cli -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py "$F" 0 | head -40
