"""Random spin kernels (0.3 ... 20 ms) on random subsets of the Trainer's side streams (optimizer, tail, weight-gradient, aux, collective) in front of
every one of eight back-to-back train steps of the small model, bf16 and fp8, three bucket sizes: the losses must equal the undisturbed run's.
The deterministic forms are tests (tests/test_model_gpu.py, test_fp8_gpu.py, test_ddp_gpu.py); this is the wider net (NOTES_r6 section 14:
36 trials, all bit-identical).  usage: python tools/fuzz_stream_delays.py [trials = 36] [seed = 1234]"""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mic_amd
from mic_amd import Trainer, create_learning_rate_fn, ops
from util_small import make_pair, batch
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
refs = {}
def run(gemm, bucket_mb, plan):
    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.1, d_layers=4, v_layers=3, **(
        dict(d_model=256, d_ffn=512, d_heads=4, v_hidden=256, v_ffn=512, v_heads=4) if gemm == "fp8" else {}))
    tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 2e-3), gemm_dtype="fp8" if gemm == "fp8" else None, bucket_mb=bucket_mb)
    r = tr.reducer
    streams = {"optimizer": r.opt_stream, "tail": r.tail_stream or r.opt_stream, "dw": ops.role_stream(dev, "dw"), "aux": ops.role_stream(dev, "aux"),
               "collective": r.stream}
    a, b = torch.zeros(1 << 18, device=dev), torch.zeros(1 << 18, device=dev)
    batches = []
    for s in range(3):
        px, labels, mask, dec_in = batch(rc, 3, 12, seed=9 + s)
        batches.append({"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()})
    losses = []
    for step in range(8):
        for (name, us) in (plan[step] if plan else []):
            with torch.cuda.stream(streams[name]):
                ops.comm_emulate(a, b, 1 << 20, us, 8)
        losses.append(tr.train_step(batches[step % 3])["loss"])
    torch.cuda.synchronize()
    return [float(x) for x in losses]
bad = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 36):
    gemm = rng.choice(["bf16", "fp8"]); bucket_mb = rng.choice([64.0, 0.25, 1.0])
    key = (gemm, bucket_mb)
    if key not in refs:
        refs[key] = run(gemm, bucket_mb, None)
    plan = [[(n, rng.choice([300.0, 2000.0, 8000.0, 20000.0])) for n in ("optimizer", "tail", "dw", "aux", "collective") if rng.random() < 0.5] for _ in range(8)]
    got = run(gemm, bucket_mb, plan)
    d = max(abs(x - y) for x, y in zip(got, refs[key]))
    ok = d <= 2e-4 * abs(refs[key][0])
    bad += not ok
    print(f"trial {trial:2d} {gemm} bucket {bucket_mb}: max |dloss| {d:.2e} {'ok' if ok else 'MISMATCH ' + str(plan)}", flush=True)
print("mismatches:", bad)
