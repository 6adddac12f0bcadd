run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stage_ms'].items()})"; }
run HP_WFA2_HCAP_LOG2=11
run HP_WFA2_HCAP_LOG2=10
run HP_WFA2_HCAP_LOG2=12
