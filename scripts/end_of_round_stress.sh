for sd in 41 42 43; do timeout 200 python bench.py --seed $sd --steps 8 --no-resident --no-drop-in --cpu-seconds 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed $sd', round(d['value']), round(d['ms_per_step'],2), d['parity'])"; done
timeout 240 python bench.py --seed 44 --steps 6 --no-resident --no-drop-in --cpu-seconds 2 --spec edit_noise=0.01 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed 44 noise 1%', round(d['value']), round(d['ms_per_step'],2), d['parity'])"
timeout 240 python bench.py --seed 45 --steps 6 --no-resident --no-drop-in --cpu-seconds 2 --seq-format ascii 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed 45 ascii', round(d['value']), round(d['ms_per_step'],2), d['parity'])"
timeout 150 python scripts/wfa_stress.py 27 60 2>&1 | tail -2
timeout 150 python scripts/wfa_stress.py 28 60 2>&1 | tail -2
