run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,1) for k,v in d['stage_ms'].items()})"; }
timeout 900 python -m pytest tests/test_blocks_gpu.py tests/test_local_gpu.py tests/test_e2e_gpu.py tests/test_cpp_mirror.py tests/test_block_io_gpu.py -x -q 2>&1 | tail -3
run A=1
run HP_BLOCK_HOST_THREADS=64
HP_DEBUG=1 python bench.py --no-cpu --steps 1 --warmup 1 2>&1 | grep "^\[hp\] rows" | tail -1
