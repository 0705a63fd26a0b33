# one train figure (ms per step) and one beam-4 figure (ms per decoder step) of the library in the tree
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train ms/step', d['ms_per_step'], 'img/s', d['value'], '| decoder step ms', d.get('beam4_generate', {}).get('ms_per_decoder_step'))"
