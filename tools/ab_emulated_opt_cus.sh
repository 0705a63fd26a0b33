# emulated 8-rank exchange on one GPU: CUs of the optimizer stream's mask and of the (assumed) collective stream
A="--steps 12 --warmup 4 --no-generate --no-cpu-baseline --no-roofline --no-dense-leg --emulate-comm 0 --emulate-main 8"
g() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['emulation']['collective_stream_busy_ms_per_step'])"; }
for r in 1 2; do
  for oc in 96 64 48 128; do echo -n "MIC_OPT_CUS=$oc comm_cus=32: "; MIC_OPT_CUS=$oc python bench.py $A 2>/dev/null | g; done
  for cc in 16 64; do echo -n "MIC_OPT_CUS=96 comm_cus=$cc: "; python bench.py $A --comm-cus $cc 2>/dev/null | g; done
done
