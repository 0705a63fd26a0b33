# Same-box A/B of the tree's libmic_hip.so against the library of an EARLIER COMMIT (default HEAD~1), on one gpurun lease.
# usage (in the build container): bash tools/ab_prev_build.sh [-r <rev>] '<command printing the figure>'
#   e.g. bash tools/ab_prev_build.sh 'bash tools/bench_train_only.sh'
# Builds <rev>'s csrc/ in a scratch worktree (hipcc cross-compiles here), leaves it as multilingual-image-captioning_amd/libmic_hip_base.so
# (git-ignored; it travels to the GPU box with the tree) and runs tools/ab_lib.sh there: new / base / new / base.
# Why not a switch inside the new build: both arms of such an A/B carry whatever the change did to the code around it — round 4 shipped
# a 15 % regression for seven commits that way (profiles/r4_gemm_w4_experiment.txt §7).  The Python side of the tree is the NEW one in
# both arms: this compares kernels, so it only makes sense while the C ABI both libraries export is the one the tree's _lib.py binds.
set -e
REV=HEAD~1
if [ "$1" = "-r" ]; then REV=$2; shift 2; fi
CMD="$1"
ROOT=$(git rev-parse --show-toplevel)
P=multilingual-image-captioning_amd
WT=$(mktemp -d /tmp/mic_prev_XXXX)
git -C "$ROOT" worktree add --detach "$WT" "$REV" > /dev/null
trap 'git -C "$ROOT" worktree remove --force "$WT" > /dev/null 2>&1 || true' EXIT
make -C "$WT/$P/csrc" -j8 > "$WT/build.log" 2>&1 || { tail -20 "$WT/build.log"; exit 1; }
cp "$WT/$P/libmic_hip.so" "$ROOT/$P/libmic_hip_base.so"
make -C "$ROOT/$P/csrc" -j8 > /dev/null
echo "# base = $(git -C "$ROOT" rev-parse --short "$REV"), new = working tree at $(git -C "$ROOT" rev-parse --short HEAD)$(git -C "$ROOT" diff --quiet || echo +dirty)"
cd "$ROOT" && /usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/ab_lib.sh '$CMD'"
