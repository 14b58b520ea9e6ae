# usage: seq2_try.sh <headline workload> <also list> "<label>" ENV=...
h=$1; a=$2; lab=$3; shift; shift; shift
env "$@" timeout 500 python bench.py --workload $h --also $a --no-cpu-baseline --no-probe --steps 20 --warmup 5 --full-out /tmp/seq_full.json 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lab', ' '.join('%s=%.4g' % (k, v['value']) for k, v in d['summary'].items()))"
