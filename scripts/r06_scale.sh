cd $GRAFT_REPO_ROOT
export FALCON_AMD_CHAIN_PASS_A=1
for n in 3072 1536 768 384 192; do
python bench.py --no-pipeline --steps 4 --warmup 1 --no-cpu-baseline --no-end-to-end --piles $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print($n, ' '.join('%s %.3f'%(a[2:],b) for a,b in k.items()), ' chain us/pile %.3f'%(1000*k['k_chain']/$n))"
done
