#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ...   -> train ms/step and beam-4 captions/s for each value of the env var, twice
VAR=$1; shift
for r in 1 2; do for v in "$@"; do echo -n "$VAR=$v train "; env $VAR=$v python bench.py --no-cpu-baseline --no-generate --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done; done
for v in "$@"; do echo -n "$VAR=$v gen "; env $VAR=$v python bench.py --generate-only 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_decoder_step'])"; done
