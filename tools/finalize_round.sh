# everything the round's profiles/ and README numbers come from, in one GPU-box call:  bash tools/finalize_round.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -rP 2>&1 | grep -E "^\[|passed|failed|skipped|Error" | grep -v "Gloo\|W1002\|socket" > $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
WINDOW=50 WINDOWS=12 python tools/soak_sustained.py > $O/r5_soak_sustained.txt 2>/dev/null
python tools/gemm_shapes_bench.py > $O/r5_gemm_vs_library_train.txt 2>/dev/null
python tools/gemm_shapes_bench.py --decode > $O/r5_gemm_vs_library_decode.txt 2>/dev/null
MD=2432 python tools/gemm_shapes_bench.py --no-head > $O/r5_gemm_vs_library_packed.txt 2>/dev/null
bash tools/profile_round.sh > $O/profile_round.log 2>&1
bash tools/profile_emulated_comm.sh r5emu > $O/profile_emu.log 2>&1
tail -3 $O/pytest_gpu.log; tail -c 600 $O/bench.json; cat $O/r5_soak_sustained.txt
