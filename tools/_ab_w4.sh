B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-generate --emulate-comm 0"
G="python bench.py --generate-only --no-cpu-baseline --no-roofline"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], (d.get('dense_captions') or {}).get('ms_per_step'))"; }
gs() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('beam4_generate') or d; print(g.get('ms_per_decoder_step'), g.get('value'))"; }
for r in 1 2 3; do
  for w in 1 0; do
    echo -n "train W4=$w   "; MIC_GEMM_W4=$w $B 2>/dev/null | ms
    echo -n "gen   W4=$w   "; MIC_GEMM_W4=$w $G 2>/dev/null | gs
  done
done
