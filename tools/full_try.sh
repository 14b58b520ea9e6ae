# usage: full_try.sh "<label>" ENV=...: the driver's default sequence (no CPU baselines), prints every workload's value
lab=$1; shift
env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --full-out /tmp/full_try.json 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lab', ' '.join('%s=%.4g' % (k, v['value']) for k, v in d['summary'].items()))"
