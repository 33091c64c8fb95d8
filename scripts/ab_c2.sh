#!/bin/bash
# A/B of the configs[1] minibatch on the GPU box: the env switches of mmg_create select the variants.
#   scripts/ab_c2.sh "VAR=1" ["VAR2=1" ...]   -> first-pass us per minibatch + HIP-event kernel averages, default build first
export MMG_BENCH_MIN_SECONDS=${MMG_BENCH_MIN_SECONDS:-0.5}
W=${W:-c2}
run() { env $1 python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print('%-22s first-pass %.2f us @ %.2f steps | window %.2f us @ %.2f steps | %s' % (sys.argv[1], 1e3*c['first_pass_ms_per_minibatch'], c['first_pass_exchange_steps_per_minibatch'], 1e3*d['ms_per_step'], c['exchange_steps_per_minibatch'], d['roofline']['kernels_us']))" "$1"; }
run "MMG_DEFAULT=1"
for v in "$@"; do run "$v"; done
