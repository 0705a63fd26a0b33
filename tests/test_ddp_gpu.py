"""Data-parallel train step, world_size 2, on the GPU box's single MI355X: both ranks share cuda:0 and exchange gradients
over gloo (RCCL needs one device per rank; the host/stream logic — bucketed all-reduce on the side stream as backward
completes the flat gradient buffer, SUM + 1/world in AdamW, metric pmean — is identical).  On a node with >= 2 GPUs
`MIC_DDP_BACKEND=nccl` runs the same tests over RCCL, one device per rank (tools/first_multi_gpu_run.sh).  Checked against the oracle:
pmean of per-rank gradients of per-rank masked-mean losses (main.py:679, 698), then one AdamW step."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world):
    """process group + this rank's device: gloo with every rank on cuda:0 (the one-GPU box), or MIC_DDP_BACKEND=nccl with one device
    per rank (RCCL; the channel cap the Trainer's CU budget is read from is exported first)"""
    import torch.distributed as dist

    if os.environ.get("MIC_DDP_BACKEND", "gloo") == "nccl":
        from mic_amd.train import configure_rccl

        configure_rccl()
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        return dev
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cuda:0")


def _worker(rank, world, port, q, comm=None):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = _init(rank, world)
    try:
        from util_small import batch, make_pair

        import mic_amd  # noqa: F401
        from mic_amd import Trainer, create_learning_rate_fn
        from mic_amd.params import flatten_tree
        from oracle import train_ref

        rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
        lr_fn = create_learning_rate_fn(40, 4, 1, 0, 1e-3)
        tr = Trainer(model, lr_fn, weight_decay=0.01, seed=42, bucket_mb=0.25,  # small buckets: several fire mid-backward
                     grad_comm_dtype=torch.bfloat16 if comm == "bf16" else None)
        gtol = 1e-2 if comm == "bf16" else 5e-4  # opt-in bf16 exchange: each rank's bucket is rounded to bf16 before the sum
        assert len(tr.buckets) > 3
        B, T = 2, 12
        shards = [batch(rc, B, T, seed=500 + r) for r in range(world)]
        px, labels, mask, dec_in = shards[rank]
        out = tr.train_step({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(),
                             "decoder_input_ids": dec_in.numpy()})
        torch.cuda.synchronize()
        ok, msg = True, ""
        if rank == 0:
            losses, grads = [], []
            for (a, b, c, d) in shards:
                l, g = train_ref.loss_and_grads(rc, p, a, b, c, d)
                losses.append(l.item())
                grads.append(g)
            gm = {k: sum(g[k] for g in grads) / world for k in p}
            if abs(float(out["loss"]) - sum(losses) / world) > 5e-5:
                ok, msg = False, f"loss {float(out['loss'])} vs {sum(losses) / world}"
            gsum = model.store.export_flat("grad")  # SUM over ranks sits in the flat buffer; 1/world is folded into AdamW
            got = flatten_tree(model.params)
            for k in p:
                sc = gm[k].abs().max().item()
                if sc > 1e-6:
                    e = ((torch.from_numpy(gsum[k]) / world - gm[k]).abs().max() / sc).item()
                    if e > gtol:
                        ok, msg = False, f"grad {k}: {e}"
                        break
                newp, _, _ = train_ref.adamw_update(p[k], gm[k], torch.zeros_like(p[k]), torch.zeros_like(p[k]), 0, 1e-3, wd=0.01)
                d = (torch.from_numpy(got[k]) - newp).abs().max().item()
                if comm is None and d > 1e-4:  # first Adam step = lr * g/(|g|+eps): 10 % of one step, dominated by near-zero gradients
                    ok, msg = False, f"param {k}: {d}"
                    break
        q.put((rank, ok, msg))
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_matches_oracle(dev):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] for r in res), res


def test_two_rank_bf16_gradient_exchange(dev):
    """Trainer(grad_comm_dtype=torch.bfloat16): same step, gradients within bf16 rounding of the fp32 mean (opt-in mode)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "bf16")) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] for r in res), res


def _fp8_worker(rank, world, port, q, lag=None):
    """configs[4] data parallel in small: two ranks, fp8 GEMMs with fused emission (from the second step on), five steps on per-rank
    batches: every rank sees the same pmean loss and ends with BIT-IDENTICAL fp32 master weights (the all-reduce and the deterministic
    embedding-row scatter are rank-symmetric: with an atomic scatter the replicas' embeddings drifted by 2.8e-9 in five steps), the
    loss falls, the producers emit the fp8 operands.  The fp8 arithmetic itself is pinned by tests/test_fp8_gpu.py."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = _init(rank, world)
    try:
        from util_small import batch, make_pair

        import mic_amd  # noqa: F401
        from mic_amd import Trainer, create_learning_rate_fn

        rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.1)
        tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 2e-3), seed=42, bucket_mb=0.25, gemm_dtype="fp8")
        assert model.engine.fp8 and model.engine.fp8_fused and len(tr.buckets) > 3
        px, labels, mask, dec_in = batch(rc, 3, 12, seed=700 + rank)
        b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
        losses = []
        side = None
        if lag is not None and rank == 1:  # ONE rank's side stream is slow (a 20-ms spin kernel in front of every step's work on it)
            from mic_amd import ops

            r = tr.reducer
            side = {"collective": r.stream, "tail": r.tail_stream or r.opt_stream, "optimizer": r.opt_stream,
                    "dw": ops.role_stream(dev, "dw"), "aux": ops.role_stream(dev, "aux")}[lag]
            sa, sb = torch.zeros(1 << 18, device=dev), torch.zeros(1 << 18, device=dev)
        for _ in range(5):
            if side is not None:
                with torch.cuda.stream(side):
                    ops.comm_emulate(sa, sb, 1 << 20, 20000.0, 8)
            losses.append(tr.train_step(b)["loss"])
        losses = [float(x) for x in losses]
        torch.cuda.synchronize()
        w = model.store.master.clone()
        ws = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        ls = [None] * world
        dist.all_gather_object(ls, losses)
        same_w = all(torch.equal(ws[0], x) for x in ws)
        maxdiff = max(float((ws[0] - x).abs().max()) for x in ws)
        where = [n for n, sg in model.store.segs.items() if any(not torch.equal(ws[0][sg.offset:sg.offset + sg.numel], x[sg.offset:sg.offset + sg.numel]) for x in ws)]
        flags = (same_w, all(l == ls[0] for l in ls), bool(np.isfinite(losses).all()), losses[-1] < losses[0] - 0.3,
                 len(model.engine._a8_ready) > 20)  # (last: the producers emitted the operands — scale histories exist)
        q.put((rank, all(flags), f"flags {flags} max weight difference {maxdiff} in {where[:8]} ready {len(model.engine._a8_ready)} losses {losses}"))
    finally:
        dist.destroy_process_group()


def test_two_rank_fp8_train_steps(dev):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fp8_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] for r in res), res


@pytest.mark.parametrize("lag", ["collective", "tail", "optimizer", "dw", "aux"])
def test_two_rank_fp8_replicas_stay_identical_when_one_ranks_side_stream_lags(dev, lag):
    """the same five data-parallel fp8 steps with ONE rank's collective / tail / optimizer / weight-gradient / aux stream delayed by a
    20-ms spin kernel per step: a hand-over somebody does not wait for would let that rank read or write something early and the
    replicas' master weights would differ"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fp8_worker, args=(r, 2, port, q, lag)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] for r in res), res


def _packed_worker(rank, world, port, q, full=False):
    """bf16, two ranks with different numbers of valid caption positions: the data-parallel step with packed decoder rows against
    the same step on padded rows (same process, same data, fresh model each): same loss, same summed gradients to the summation-order
    tolerance, on both ranks; the sparse embedding-row exchange carries a FIXED number of rows per rank whatever the rank's valid
    count."""
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dev = _init(rank, world)
    try:
        from util_small import batch, make_pair

        import mic_amd  # noqa: F401
        from mic_amd import Trainer, create_learning_rate_fn

        if full:  # ViT-B/32 + mBART-large-50 size, 16 of the bench's ragged captions per rank (different valid-row counts per rank)
            import bench
            from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration

            b = bench.synth_batch(16, 64, 250054, 224, 900 + rank)
        else:
            B, T = 3, 16
            px, labels, mask, dec_in = batch(make_pair(torch.bfloat16, dev, dropout=0.0)[0], B, T, seed=900 + rank)
            b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
        res = {}
        for pack in (True, False):
            if full:
                cfg = CLIPVisionMBartConfig(mbart_config=dict(dropout=0.0), clip_vision_config={})
                model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)  # same seed: same weights on both ranks
                tr = Trainer(model, create_learning_rate_fn(10_000_000, 32, 7, 0, 1e-4), seed=42, pack_rows=pack, grad_comm_dtype=None)
                assert len(tr.buckets) > 20
            else:
                rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
                tr = Trainer(model, create_learning_rate_fn(40, 4, 1, 0, 1e-3), seed=42, bucket_mb=0.25, pack_rows=pack)
            out = tr.train_step(b)
            torch.cuda.synchronize()
            assert (tr._pack is not None) == pack
            res[pack] = (float(out["loss"]), model.store.grad.clone(), model.store.lp.float().clone())
            del tr, model
        ok, msg = True, ""
        # reduced model: the same number (every valid row goes through the same per-row arithmetic).  Full size at 16 captions per
        # rank: the packed (~560 rows) and the padded (1024 rows) step get different K-group counts from the tile planner in FORWARD
        # GEMMs too, i.e. different fp32 summation orders: equal to bf16 rounding, not bit for bit
        if (abs(res[True][0] - res[False][0]) > 1e-4 * abs(res[False][0])) if full else (res[True][0] != res[False][0]):
            ok, msg = False, f"loss {res[True][0]} vs {res[False][0]}"
        g1, g0 = res[True][1], res[False][1]
        e = ((g1 - g0).abs().max() / g0.abs().max()).item()
        # reduced model (2 + 2 layers): summation order only.  Full size: the packed and the padded step pick different tile
        # configurations for their different row counts, whose last-bit differences in the bf16 activation gradients cascade through
        # 12 + 12 layers exactly like the forward cascade of tests/test_oracle_cpu.py — the direction must agree, not the last bits
        if e > (2e-2 if full else 2e-3):
            ok, msg = False, f"summed gradients differ: {e}"
        c = torch.nn.functional.cosine_similarity(g1.double(), g0.double(), dim=0).item()
        if c < 0.9995:
            ok, msg = False, f"summed gradients: cosine {c}"
        if rank == 0:
            print(f"[two ranks, {'full size' if full else 'reduced'}] packed vs padded: loss {res[True][0]} / {res[False][0]}, summed gradients max diff {e:.2e} of the largest entry, cosine {c:.6f}", flush=True)
        # every rank ends with the same weights (the all-reduce and the embedding-row exchange are rank-symmetric)
        lp = res[True][2]
        gathered = [torch.zeros_like(lp.cpu()) for _ in range(world)]
        dist.all_gather(gathered, lp.cpu())
        if not all(torch.equal(gathered[0], t) for t in gathered):
            ok, msg = False, "ranks ended with different weights"
        q.put((rank, ok, msg))
    finally:
        dist.destroy_process_group()


def test_two_rank_packed_rows_equal_padded_rows(dev):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert all(r[1] for r in res), res


def test_two_rank_packed_rows_equal_padded_rows_full_size(dev):
    """the same at ViT-B/32 + mBART-large-50 size: two ranks with different valid-row counts (16 ragged captions each), the bucket plan
    of the real layout (the tied embedding in eight pieces), the sparse embedding-row exchange at 1024 fixed rows per rank"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30300 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=900) for _ in range(2))
    for pr in procs:
        pr.join(timeout=120)
    assert all(r[1] for r in res), res
