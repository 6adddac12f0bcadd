"""GPU bring-up helper: runs tiny solves in subprocesses with hard timeouts and logs what hangs."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out"); os.makedirs(OUT, exist_ok=True)
CASES = {
 "devcount": "from hiphase_amd import _ffi; print('devices', _ffi.lib().hp_device_count())",
 "n1": "from hiphase_amd import *; b,_=synth_block(1,5,2,0,0,6); r=astar_solver(0,b); print(r.haplotype_1, r.statistics.as_tuple())",
 "n4": "from hiphase_amd import *; b=BlockMatrix.from_rows([([0]*4,[2]*4),([1]*4,[3]*4)]); r=astar_solver(0,b); print(r.haplotype_1, r.haplotype_2, r.statistics.as_tuple())",
 "n50": "from hiphase_amd import *; b,_=synth_block(50,8,20,0.01,0.02,1); r=astar_solver(0,b); print(r.haplotype_1, r.statistics.as_tuple())",
 "n300": "from hiphase_amd import *; b,_=synth_block(300,30,20,0.1,0.02,2); r=astar_solver(0,b); print(r.statistics.as_tuple())",
}
log = open(os.path.join(OUT, "debug.log"), "w")
runs = []
for a in (sys.argv[1:] or list(CASES)):
    name, _, stage = a.partition(":")
    runs.append((name, stage))
for name, stage in runs:
    t = time.time()
    env = dict(os.environ, HP_DEBUG="1")
    if stage: env["HP_DEBUG_STAGE"] = stage
    try:
        p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,%r); sys.path.insert(0,%r+'/tests')\n" % (ROOT, ROOT) + CASES[name]],
                           capture_output=True, text=True, timeout=25, env=env)
        msg = f"[{name}:{stage}] rc={p.returncode} {time.time()-t:.1f}s\nSTDOUT: {p.stdout[-2000:]}\nSTDERR: {p.stderr[-2000:]}\n"
    except subprocess.TimeoutExpired as e:
        msg = f"[{name}:{stage}] TIMEOUT\nSTDOUT: {(e.stdout or b'')[-2000:]}\nSTDERR: {(e.stderr or b'')[-2000:]}\n"
    print(msg, flush=True); log.write(msg); log.flush()
