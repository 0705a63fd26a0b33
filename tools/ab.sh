#!/bin/bash
# same-box A/B: ab/prev (committed) vs working tree
python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do for t in ab/prev .; do (cd $t; echo -n "$t train "; python bench.py --no-cpu-baseline --no-generate --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); done; done
for t in ab/prev .; do (cd $t; echo -n "$t gen "; python bench.py --generate-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_decoder_step'])"; python tools/bench_gemm_k2.py 2>&1 | grep "M=4096 N=4096" | grep "K=  4096\|K= 16384"); done
