"""Would two independent half-batch chains on two streams fill the chip better than one full-batch chain?  (The one-round GEMM
launches of the packed train step run on 168 of 256 CUs and are latency-bound; LayerNorm / attention kernels are latency-bound
too.)  Times forward + backward (no optimizer) of the full-size model: one Engine on the whole batch against two Engines that
share the parameter store (their gradient writes race: timing only) on half a batch each, issued A.fwd, B.fwd, A.bwd, B.bwd."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mic_amd  # noqa: F401
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, loss_rows, ops, packed_rows
from mic_amd.engine import Engine

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import synth_batch  # noqa: E402

dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config={}, clip_vision_config={})
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, seed=0, dtype=torch.bfloat16, device=dev)
st = model.store
st.ensure_grads()
B, T = 64, 64


def prep(b):
    Bh = b["input_ids"].shape[0]
    idx, rl = loss_rows(b["attention_mask"], b["input_ids"])
    q_off, q_len, ids_p, pos_p = packed_rows(b["attention_mask"], b["decoder_input_ids"])
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    return dict(px=d(b["pixel_values"], torch.float32), ids=d(ids_p, torch.int32), pos=d(pos_p, torch.int32), labels=d(b["input_ids"], torch.int32).reshape(-1),
                rows=(d(idx, torch.int32), len(idx)), rl=d(rl, torch.int32), pack=(d(q_off, torch.int32), d(q_len, torch.int32), len(idx)), B=Bh)


full = synth_batch(B, T, 250054, 224, 1234)
halves = [{k: v[i * B // 2:(i + 1) * B // 2] for k, v in full.items()} for i in range(2)]
pf, ph = prep(full), [prep(h) for h in halves]
engA, engB = model.engine, Engine(st)
for e in (engA, engB):
    e.dw_overlap = False


def run(eng, p, seed):
    return eng.loss_and_grads(p["px"], p["ids"], p["pos"], None, p["labels"], p["B"], T, seed=seed, rows=p["rows"], row_labels=p["rl"], pack=p["pack"])


def fwd(eng, p, seed):
    """forward + loss + dlogits only"""
    P = eng.P
    _, ehs = eng.vit_forward(p["px"], True)
    hf = eng.decoder_forward(p["ids"], p["pos"], None, ehs, p["B"], T, True, seed, pack=p["pack"])
    logits, stat = eng.compact_head(hf, p["B"] * T, p["rows"], packed=True)
    eng.loss_and_dlogits(logits, p["rl"], eng.ones_i32(p["rows"][1]), p["rows"][1], 0.0, backward=True, stat=stat)
    return ehs, logits


def bwd(eng, p, seed, ehs, logits):
    dehs = eng.decoder_backward(p["B"], T, p["ids"], p["pos"], None, ehs, logits, seed, rows=p["rows"], pack=p["pack"])
    eng.vit_backward(p["B"], dehs)
    eng.dw_join()


def one_chain(n):
    with ops.pinned_stream():
        for i in range(n):
            ops.zero(st.grad[st.atomic_begin:])
            e, l = fwd(engA, pf, 7 + i)
            bwd(engA, pf, 7 + i, e, l)


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def two_chains(n):
    for i in range(n):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        outs = []
        for eng, p, s in ((engA, ph[0], sA), (engB, ph[1], sB)):
            s.wait_event(ev)
            with torch.cuda.stream(s), ops.pinned_stream():
                outs.append(fwd(eng, p, 7 + i))
        for (eng, p, s), (e, l) in zip(((engA, ph[0], sA), (engB, ph[1], sB)), outs):
            with torch.cuda.stream(s), ops.pinned_stream():
                bwd(eng, p, 7 + i, e, l)
        for s in (sA, sB):
            torch.cuda.current_stream().wait_stream(s)


def two_chains_threads(n):
    """the same two chains, each issued by its own host thread (ctypes releases the GIL inside the launches, so the two launch
    sequences interleave kernel by kernel instead of phase by phase)"""
    import threading

    def work(eng, p, s):
        with torch.cuda.stream(s):
            for i in range(n):
                e, l = fwd(eng, p, 7 + i)
                bwd(eng, p, 7 + i, e, l)

    ts = [threading.Thread(target=work, args=a) for a in ((engA, ph[0], sA), (engB, ph[1], sB))]
    for s in (sA, sB):
        s.wait_stream(torch.cuda.current_stream())
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for s in (sA, sB):
        torch.cuda.current_stream().wait_stream(s)


for name, fn in (("one chain, batch 64", one_chain), ("two chains, 2 x batch 32", two_chains), ("two chains, two host threads", two_chains_threads),
                 ("one chain, batch 64", one_chain), ("two chains, 2 x batch 32", two_chains), ("two chains, two host threads", two_chains_threads)):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(8)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per forward + backward (no optimizer)")
