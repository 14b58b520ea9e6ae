# usage: seq_try.sh "<label>" ENV=...: the headline followed by ffp15 in ONE process (as in the default sequence), prints both values
lab=$1; shift
env "$@" timeout 400 python bench.py --workload cascade --also ffp15 --no-cpu-baseline --no-probe --steps 20 --warmup 5 --full-out /tmp/seq_full.json 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lab', 'cascade', round(d['value'],1), 'ffp15', round(d['summary']['ffp15']['value'],1))"
