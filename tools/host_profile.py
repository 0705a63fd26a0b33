"""Host-side cost of one train step: (a) full-size layer count on a tiny-width model (GPU time negligible -> pure Python/ctypes
enqueue time), (b) cProfile of that step."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mic_amd, bench
from mic_amd import CLIPVisionMBartConfig, FlaxCLIPVisionMBartForConditionalGeneration, Trainer, create_learning_rate_fn, loss_rows
dev = torch.device("cuda:0")
cfg = CLIPVisionMBartConfig(mbart_config=dict(vocab_size=5003, d_model=128, decoder_layers=12, decoder_attention_heads=2, decoder_ffn_dim=256),
                            clip_vision_config=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=12, num_attention_heads=2, image_size=64, patch_size=32))
model = FlaxCLIPVisionMBartForConditionalGeneration(cfg, dtype=torch.bfloat16, device=dev)
tr = Trainer(model, create_learning_rate_fn(10**7, 8, 7, 1000, 5e-5))
b = bench.synth_batch(8, 64, 5003, 64, 1)
db = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
idx, rl = loss_rows(b["attention_mask"], b["input_ids"]); db["loss_rows"] = (torch.from_numpy(idx).to(dev), torch.from_numpy(rl).to(dev))
for _ in range(3): tr.train_step(db)
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.train_step(db); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"12+12-layer tiny-width model: host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.train_step(db)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
