cd $GRAFT_REPO_ROOT
python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d['ops'].items(): print(k, v)"
