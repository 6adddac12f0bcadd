run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stage_ms'].items()})"; }
timeout 600 python -m pytest tests/test_wfa_gpu.py tests/test_blocks_gpu.py tests/test_coalesce_gpu.py -x -q 2>&1 | tail -2
HP_DEBUG=1 timeout 100 python bench.py --no-cpu --steps 1 --warmup 1 2>&1 | grep "^\[hp\] wfa2: 1280\|rows:" | tail -2 | cut -c1-330
run A=1
run HP_WFA2_TWO_PHASE=0
