# kernel traces of the train step: default, and with the gradient exchange among 8 / 2 ranks EMULATED on this GPU (bench.py --emulate-main)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r4emu}; mkdir -p $O
A="--steps 3 --warmup 2 --no-generate --no-cpu-baseline --no-dense-leg --no-extra-legs --no-roofline --emulate-comm 0"
rocprofv3 --kernel-trace --output-format csv -d $O/base -- python3 $R/bench.py $A > $O/base.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/emu8 -- python3 $R/bench.py $A --emulate-main 8 > $O/emu8.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/emu2 -- python3 $R/bench.py $A --emulate-main 2 > $O/emu2.log 2>&1
cd $R
for k in base emu8 emu2; do python tools/step_timeline.py $(ls $O/$k/*/*kernel_trace.csv | head -1) 1 1 > $O/timeline_$k.txt; done
rm -rf $O/base $O/emu8 $O/emu2
