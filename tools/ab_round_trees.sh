# whole-tree same-box A/B: the tree of an earlier round (a git worktree under ab/, built there; ab/ is git-ignored but travels to the GPU
# box) against this tree — both legs of bench.py, interleaved.   usage (on the GPU box): bash tools/ab_round_trees.sh ab/r4 [pairs=2]
OLD=${1:-ab/r4}; PAIRS=${2:-2}
leg() {  # tree dir
  X=$(grep -q -- --no-extra-legs $1/bench.py && echo --no-extra-legs)  # (older trees have no such flag, and no extra legs)
  (cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dense-leg $X --emulate-comm 0 2>/dev/null | grep '^{' | head -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['beam4_generate']; print(d['ms_per_step'], 'ms', d['value'], 'images/s |', b['ms_per_decoder_step'], 'ms', b['value'], 'captions/s')")
}
for i in $(seq $PAIRS); do echo "old ($OLD): $(leg $OLD)"; echo "new (.):     $(leg .)"; done
