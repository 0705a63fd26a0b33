#!/bin/bash
# First lease on a node with N >= 2 MI355X: everything needed to turn "RCCL has never run" into a SCALE line instead of a debugging
# session.  Run from the repo root on the node:   bash tools/first_multi_gpu_run.sh [N=all visible GPUs]
# Writes gpurun_out/multi_gpu/{env.txt, rccl_probe.log, ddp_tests.log, bench_N*.json, summary.txt}.  Every step is a plain child
# process started before anything in THIS shell touches a GPU (no exec from a GPU-initialised process), and is bounded by `timeout`.
set -u
N=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
O=gpurun_out/multi_gpu; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
{ echo "GPUs visible: $N"; /opt/rocm/bin/rocm-smi --showtopo 2>/dev/null | head -60; env | grep -E "NCCL|RCCL|HSA_|HIP_|ROCR" ; } > $O/env.txt 2>&1
[ "$N" -ge 2 ] || { echo "needs >= 2 GPUs (sees $N)" | tee $O/summary.txt; exit 2; }

# 1. RCCL comes up: an all-reduce probe over N ranks with RCCL's own INIT log, the channel cap exported as the Trainer will see it
cat > $O/_probe.py <<'PY'
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import mic_amd
from mic_amd.train import configure_rccl, allreduce_ms
cap = configure_rccl()
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(r); dev = torch.device("cuda", r)
dist.init_process_group("nccl", rank=r, world_size=w, device_id=dev)
x = torch.ones(1, device=dev); dist.all_reduce(x); torch.cuda.synchronize(); assert int(x.item()) == w
for mb in (64, 128, 1024):   # the bucket sizes of the step: 64-128 MB pieces, and the whole embedding for comparison
    t = torch.ones(mb * 262144, device=dev)
    for _ in range(3): dist.all_reduce(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dist.all_reduce(t)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    if r == 0:
        from mic_amd.train import allreduce_projections
        print(f"all-reduce {mb} MB fp32 over {w} ranks: {ms:.2f} ms measured; projected {allreduce_projections(mb * 1048576.0, w)} (the link rate that fits goes into choose_comm_dtype(link_gbps=)), cap {cap} channels", flush=True)
dist.destroy_process_group()
PY
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 $O/_probe.py > $O/rccl_probe.log 2>&1
echo "rccl probe: rc $?" | tee $O/summary.txt
grep -E "all-reduce|coll channels|Channel 00/" $O/rccl_probe.log | head -8 | tee -a $O/summary.txt

# 2. the data-parallel tests of tests/test_ddp_gpu.py over RCCL, one device per rank: two ranks against the oracle's mean of per-rank
#    gradients (fp32 and bf16 exchange), packed = padded at the reduced and at the full size, both ranks end with the same weights
MIC_DDP_BACKEND=nccl timeout 1800 python -m pytest tests/test_ddp_gpu.py -q -m gpu -x > $O/ddp_tests.log 2>&1
echo "ddp tests over RCCL: rc $? ($(tail -1 $O/ddp_tests.log))" | tee -a $O/summary.txt

# 3. the scaling line: weak scaling, 64 images per GPU, fp32 exchange (the reference's pmean), then the bf16 opt-in at the sizes where
#    the byte counts say fp32 cannot hide (N = 2, 4)
for n in 1 2 4 8; do
  [ $n -le $N ] || continue
  timeout 1200 python bench.py --gpus $n --steps 20 --warmup 5 --no-generate --no-cpu-baseline --no-fp8-leg --emulate-comm 0 > $O/bench_N$n.json 2> $O/bench_N$n.err
  echo "bench --gpus $n (fp32 exchange): rc $? $(grep '^{' $O/bench_N$n.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "images/s", d["ms_per_step"], "ms", d["config"]["grad_allreduce"])' 2>/dev/null)" | tee -a $O/summary.txt
  if [ $n -eq 2 ] || [ $n -eq 4 ]; then
    timeout 1200 python bench.py --gpus $n --steps 20 --warmup 5 --grad-comm bf16 --no-generate --no-cpu-baseline --emulate-comm 0 > $O/bench_N${n}_bf16.json 2> $O/bench_N${n}_bf16.err
    echo "bench --gpus $n (bf16 exchange): rc $? $(grep '^{' $O/bench_N${n}_bf16.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "images/s", d["ms_per_step"], "ms")' 2>/dev/null)" | tee -a $O/summary.txt
  fi
done
# 4. configs[4] on the node: the fp8 step (QKV / FFN projections and the tied LM head as fp8 GEMMs, fused emission) at every N, fp32 exchange
for n in 1 2 4 8; do
  [ $n -le $N ] || continue
  timeout 1200 python bench.py --gpus $n --dtype fp8 --steps 20 --warmup 5 --no-generate --no-cpu-baseline --no-extra-legs --emulate-comm 0 > $O/bench_fp8_N$n.json 2> $O/bench_fp8_N$n.err
  echo "bench --gpus $n --dtype fp8: rc $? $(grep '^{' $O/bench_fp8_N$n.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "images/s", d["ms_per_step"], "ms")' 2>/dev/null)" | tee -a $O/summary.txt
done
echo "done: $O/summary.txt"
