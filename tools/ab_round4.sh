# same-box A/Bs of round 4 (train ms per step; two interleaved runs per arm)
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-generate --emulate-comm 0"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], (d.get('dense_captions') or {}).get('ms_per_step'))"; }
for r in 1 2; do
  echo -n "default            "; $B 2>/dev/null | ms
  echo -n "MIC_GEMM_T192=0    "; MIC_GEMM_T192=0 $B 2>/dev/null | ms
  echo -n "MIC_GEMM_QUANT=1   "; MIC_GEMM_QUANT=1 $B 2>/dev/null | ms
done
