#!/bin/bash
# A/B of one environment switch on one time_configs case: scripts/env_sweep.sh <case prefix> <VAR> <value> [<value> ...]
# (three runs of 400 minibatches per value; the library reads its switches at mmg_create)
C=$1; VAR=$2; shift 2
for v in "$@"; do
  for rep in 1 2 3; do
    echo -n "$VAR=$v  "; env $VAR=$v N=400 python "$(dirname "$0")/time_configs.py" "$C" 2>&1 | grep "^$C" | cut -c1-110
  done
done
