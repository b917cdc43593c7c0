cd $GRAFT_REPO_ROOT
S="--workload c3 --genomes 116 --families 1 --queries 64 --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive-step"
run() { tag=$1; shift; env "$@" timeout 120 python bench.py $S --tag $tag 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$tag', p['value'], p['ms_per_step'], p['rows'], {k: round(v) for k,v in p['stage_ms'].items()})
print('   ', [(k['name'], k['launches'], k['avg_ms']) for k in p['kernels'][:8]])
"; }
run default
run nowin LM_WFA_WIN=00001
run segwave LM_PA_SEG_BY_WAVE=1
run lanes LM_CHAIN1_LANES=1
