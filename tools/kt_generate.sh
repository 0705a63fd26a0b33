# kernel trace of the beam-4 leg -> gpurun_out/$1/generate_kernel_stats.txt   (usage: bash tools/kt_generate.sh <tag>)
R=$GRAFT_REPO_ROOT; TAG=${1:-ktg}; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --generate-only --no-roofline > $O/kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 378 > $O/generate_kernel_stats.txt
rm -rf $O/kt
head -22 $O/generate_kernel_stats.txt
