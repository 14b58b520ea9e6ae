# usage: wl_try.sh <workload> "<label>" ENV=... : one bench run of a workload with the given environment, prints value and ms/step
wl=$1; lab=$2; shift; shift
env "$@" timeout 300 python bench.py --workload $wl --also none --no-cpu-baseline --no-probe --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$wl $lab',d['value'],d['ms_per_step'])"
