# everything the round's profiles/ and README numbers come from, in one GPU-box call:  bash tools/finalize_round.sh
R=$GRAFT_REPO_ROOT; RD=${ROUND:-r6}; export ROUND=$RD; O=$R/gpurun_out/${RD}final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[|passed|failed|skipped|Error" | grep -v "Gloo\|W1002\|socket" > $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
# same-box A/B of the two GEMM modes (two interleaved pairs)
for i in 1 2; do for d in bf16 fp8; do echo "$d $(python bench.py --dtype $d --steps 20 --warmup 5 --no-generate --no-cpu-baseline --no-roofline --no-dense-leg --no-extra-legs --emulate-comm 0 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "images/s")')"; done; done > $O/${RD}_ab_bf16_vs_fp8.txt
python tools/h2d_probe.py > $O/${RD}_h2d_probe.txt 2>/dev/null
WINDOW=50 WINDOWS=12 python tools/soak_sustained.py > $O/${RD}_soak_sustained.txt 2>/dev/null
python tools/gemm_shapes_bench.py > $O/${RD}_gemm_vs_library_train.txt 2>/dev/null
python tools/gemm_shapes_bench.py --decode > $O/${RD}_gemm_vs_library_decode.txt 2>/dev/null
MD=2432 python tools/gemm_shapes_bench.py --no-head > $O/${RD}_gemm_vs_library_packed.txt 2>/dev/null
bash tools/profile_round.sh > $O/profile_round.log 2>&1
bash tools/profile_emulated_comm.sh ${RD}emu > $O/profile_emu.log 2>&1
tail -3 $O/pytest_gpu.log; tail -c 600 $O/bench.json; cat $O/${RD}_soak_sustained.txt
