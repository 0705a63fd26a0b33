# prints "ms_per_step images/s" of the train leg only (20 timed steps): the figure the same-box A/B scripts compare
python bench.py --steps ${STEPS:-20} --warmup 5 --no-generate --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --emulate-comm 0 2>/dev/null | grep '^{' | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
