# round-5 A/B of how the weight-gradient stream shares the chip with backward (same box, one bench run per arm, base first and last)
# usage: bash tools/ab_overlap_r5.sh   -> gpurun_out/r5ab/overlap.txt
mkdir -p gpurun_out/r5ab
OUT=gpurun_out/r5ab/overlap.txt
: > $OUT
run() {  # label, env assignments...
  label="$1"; shift
  line=$(env "$@" python bench.py --steps 20 --warmup 5 --no-generate --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --emulate-comm 0 2>/dev/null | grep '^{' | head -1)
  echo "$label $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['ms_per_step'], d['value'])" "$line")" | tee -a $OUT
}
run base            MIC_NOP=1
run prio_step_hi    MIC_PRIO_STEP=-1
run dw_cus128       MIC_DW_CUS=128
run dw_cus96        MIC_DW_CUS=96
run dw_cus160       MIC_DW_CUS=160
run dw128_free192   MIC_DW_CUS=128 MIC_FREE_CUS=192
run dw128_prio      MIC_DW_CUS=128 MIC_PRIO_STEP=-1
run dw_off          MIC_DW_OVERLAP=0
run base2           MIC_NOP=1
