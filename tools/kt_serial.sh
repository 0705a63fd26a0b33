# serial kernel trace of the train step (nothing overlapped) -> gpurun_out/$1/train_kernel_stats_serial.txt   (usage: bash tools/kt_serial.sh <tag> [extra env...])
R=$GRAFT_REPO_ROOT; TAG=${1:-kt}; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
export MIC_DW_OVERLAP=0 MIC_OPT_OVERLAP=0
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0 > $O/kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 4 > $O/train_kernel_stats_serial.txt
rm -rf $O/kt
head -30 $O/train_kernel_stats_serial.txt
