"""Can RCCL run two ranks on ONE device?  (A gpurun box has one GPU; NCCL refuses duplicate devices.)  Run under torchrun:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_same_gpu_probe.py
Prints the outcome; never hangs longer than its own timeout."""
import datetime
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]), timeout=datetime.timedelta(seconds=60),
                            device_id=torch.device("cuda", 0))
    x = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f"rank {rank}: RCCL all_reduce on a shared device OK -> {x[0].item()}")
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL on a shared device failed: {type(e).__name__}: {str(e)[:300]}")
