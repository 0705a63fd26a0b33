# every profiling pass of a round (usage: ROUND=r6 bash tools/profile_round.sh); rocprofv3 may segfault at EXIT on this pool: its CSVs are complete by then
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
RD=${ROUND:-r6}
O=$R/gpurun_out/${RD}prof; mkdir -p $O
TR="--steps 1 --warmup 1 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0"
# train: kernel trace with the default overlap (dW / optimizer streams) and serial
rocprofv3 --kernel-trace --output-format csv -d $O/train_kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0 > $O/train_kt.log 2>&1
export MIC_DW_OVERLAP=0 MIC_OPT_OVERLAP=0
rocprofv3 --kernel-trace --output-format csv -d $O/train_kt_serial -- python3 $R/bench.py --steps 3 --warmup 1 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0 > $O/train_kt_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/train_fetch -- python3 $R/bench.py $TR > $O/train_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/train_write -- python3 $R/bench.py $TR > $O/train_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/train_mfma -- python3 $R/bench.py $TR > $O/train_mfma.log 2>&1
# configs[4]: the fp8 step, serial (the quantiser launch count of a steady step: 8 steps traced, the first one has no scale history)
rocprofv3 --kernel-trace --output-format csv -d $O/fp8_kt_serial -- python3 $R/bench.py --dtype fp8 --steps 6 --warmup 2 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0 > $O/fp8_kt_serial.log 2>&1
unset MIC_DW_OVERLAP MIC_OPT_OVERLAP
# generate
rocprofv3 --kernel-trace --output-format csv -d $O/gen_kt -- python3 $R/bench.py --generate-only --no-roofline > $O/gen_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/gen_fetch -- python3 $R/bench.py --generate-only --no-roofline > $O/gen_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/gen_write -- python3 $R/bench.py --generate-only --no-roofline > $O/gen_write.log 2>&1
cd $R
f() { ls $O/$1/*/*$2 | head -1; }
python tools/rocpd_stats.py $(f train_kt kernel_trace.csv) 4 > $O/${RD}_train_kernel_stats_overlapped.txt
python tools/rocpd_stats.py $(f train_kt_serial kernel_trace.csv) 4 > $O/${RD}_train_kernel_stats_serial.txt
python tools/rocpd_stats.py $(f fp8_kt_serial kernel_trace.csv) 8 > $O/${RD}_train_fp8_kernel_stats_serial.txt
python tools/step_timeline.py $(f train_kt kernel_trace.csv) 1 1 > $O/${RD}_train_timeline.txt
python tools/pmc_traffic.py $(f train_fetch counter_collection.csv) $(f train_write counter_collection.csv) 2 $O/${RD}_train_pmc_hbm_traffic.json > $O/${RD}_train_pmc_hbm_traffic.txt
python tools/pmc_mfma.py $(f train_mfma counter_collection.csv) > $O/${RD}_train_pmc_mfma_lds.txt
python tools/rocpd_stats.py $(f gen_kt kernel_trace.csv) 378 > $O/${RD}_generate_kernel_stats.txt
python tools/pmc_traffic_gen.py $(f gen_fetch counter_collection.csv) $(f gen_write counter_collection.csv) 378 $O/${RD}_generate_pmc_hbm_traffic.json > $O/${RD}_generate_pmc_hbm_traffic.txt
# keep only the summaries and the kernel-trace csv of the two main runs (the merge-back limit is 64 MiB)
cp $(f train_kt_serial kernel_trace.csv) $O/${RD}_train_rocprofv3_kernel_trace_serial.csv
cp $(f gen_kt kernel_trace.csv) $O/${RD}_generate_rocprofv3_kernel_trace.csv
rm -rf $O/fp8_kt_serial $O/train_kt $O/train_kt_serial $O/train_fetch $O/train_write $O/train_mfma $O/gen_kt $O/gen_fetch $O/gen_write
ls -la $O; head -5 $O/${RD}_train_pmc_hbm_traffic.txt; cat $O/${RD}_train_pmc_hbm_traffic.json; cat $O/${RD}_generate_pmc_hbm_traffic.json; head -12 $O/${RD}_train_pmc_mfma_lds.txt
