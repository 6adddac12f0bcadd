run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stage_ms'].items()})"; }
timeout 900 python -m pytest tests/test_wfa_gpu.py tests/test_blocks_gpu.py tests/test_coalesce_gpu.py -x -q 2>&1 | tail -3
run A=1
run A=2
