A="--steps 10 --warmup 3 --no-generate --no-cpu-baseline --no-roofline --no-dense-leg"
g() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('comm_emulated'); print(d['ms_per_step'], {k:(v['ms_per_step'], v['collective_stream_busy_ms_per_step']) for k,v in e['worlds'].items()} if e else '')"; }
echo -n "main-emulated 8:       "; python bench.py $A --emulate-comm 0 --emulate-main 8 2>/dev/null | g
echo -n "legs 8,8,2:            "; python bench.py $A --emulate-comm 8,8,2 2>/dev/null | g
echo -n "legs 8,8,2 HWQ=8:      "; GPU_MAX_HW_QUEUES=8 python bench.py $A --emulate-comm 8,8,2 2>/dev/null | g
echo -n "main-emulated 8 HWQ=8: "; GPU_MAX_HW_QUEUES=8 python bench.py $A --emulate-comm 0 --emulate-main 8 2>/dev/null | g
echo -n "plain HWQ=8:           "; GPU_MAX_HW_QUEUES=8 python bench.py $A --emulate-comm 0 2>/dev/null | g
echo -n "plain:                 "; python bench.py $A --emulate-comm 0 2>/dev/null | g
