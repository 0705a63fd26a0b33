import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from util_small import batch, make_pair
from oracle import model_ref as M
dev = torch.device('cuda:0')
rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
B, K, L = 2, 4, 8
px, *_ = batch(rc, B, 12, seed=1)
with torch.no_grad():
    ehs, _ = M.encode(rc, p, px)
enc = model.encode(px.numpy())
R = B * K
cache = model.init_cache(R, L)
src = torch.zeros((R, L), dtype=torch.int32, device=dev); src[:, 0] = torch.arange(R, dtype=torch.int32)
cache["src_row"] = src
st = model.store
model._decode_set_encoder(cache, enc.last_hidden_state.reshape(B * st.S, st.d), B, K)
state = M.DecodeState(rc, R, L)
tok = torch.full((R,), 2, dtype=torch.int32, device=dev)
pos = torch.zeros(R, dtype=torch.int32, device=dev)
logits = model._decode_step(cache, tok, pos)
torch.cuda.synchronize()
with torch.no_grad():
    ref = M.decode_step(rc, p, state, torch.full((R, 1), 2), torch.zeros((R, 1), dtype=torch.int64), ehs.repeat_interleave(K, 0))
got = logits[:R, :rc.vocab_size].cpu()
print("nan:", torch.isnan(got).sum().item(), "err", (got - ref[:, 0]).abs().max().item())
for r in range(R):
    print(r, (got[r] - ref[r, 0]).abs().max().item())
