cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fp8_r2d -- python3 bench.py --dtype fp8 --steps 3 --warmup 1 --no-generate --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 tools/rocpd_stats.py gpurun_out/prof_fp8_r2d/*/*_kernel_trace.csv 4 | head -16 | cut -c1-150
python3 tools/rocpd_stats.py gpurun_out/prof_fp8_r2d/*/*_kernel_trace.csv 4 | grep -E "fp8_|colsum"
