run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stage_ms'].items()})"; }
timeout 600 python -m pytest tests/test_wfa_gpu.py -x -q 2>&1 | tail -2
timeout 250 python scripts/bench_wfa.py --jobs 65536 --cpu-sample 1 --kernel compact 2>&1 | tail -1 | cut -c1-160
HP_DEBUG=1 timeout 100 python bench.py --no-cpu --steps 1 --warmup 1 2>&1 | grep "^\[hp\] wfa2: 1280\|handed back" | tail -4 | cut -c1-230
run A=1
