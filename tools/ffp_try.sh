# usage: ffp_try.sh "<label>" ENV=... : one ffp15 bench run with the given environment, prints value and ms/step
lab=$1; shift
env "$@" timeout 200 python bench.py --workload ffp15 --also none --no-cpu-baseline --no-probe --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lab',d['value'],d['ms_per_step'])"
