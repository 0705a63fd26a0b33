import os, sys
os.environ["HIP_LAUNCH_BLOCKING"] = "1"
os.environ["AMD_SERIALIZE_KERNEL"] = "3"
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from util_small import batch, make_pair
from mic_amd import ops
import mic_amd.ops as O
dev = torch.device('cuda:0')
rc, p, model = make_pair(torch.float32, dev, gelu="tanh", decoder_ln_eps=1e-6)
px, labels, mask, dec_in = batch(rc, 2, 12, seed=1)
# wrap every op with a sync + print
for name in dir(O):
    f = getattr(O, name)
    if callable(f) and not name.startswith('_') and name not in ('Optional',):
        def mk(f, name):
            def w(*a, **k):
                r = f(*a, **k); torch.cuda.synchronize(); return r
            return w
        if f.__module__ == O.__name__:
            setattr(O, name, mk(f, name))
import faulthandler; faulthandler.enable()
print("greedy..."); sys.stdout.flush()
out = model.generate(px.numpy(), num_beams=1, max_length=10)
print(out.sequences.cpu()); sys.stdout.flush()
print("beam..."); sys.stdout.flush()
_bs = O.beam_step
def dbg_bs(B, K, max_len, V, cur_len, eos, pad, lp, es, cand_val, cand_idx, running_seq, running_scores, seq, scores, finished, src_row, next_token, flags):
    print("cur_len", cur_len, "cand_val", cand_val.cpu()[:4], "cand_idx", cand_idx.cpu()[:4]); sys.stdout.flush()
    _bs(B, K, max_len, V, cur_len, eos, pad, lp, es, cand_val, cand_idx, running_seq, running_scores, seq, scores, finished, src_row, next_token, flags)
    print(" next_token", next_token.cpu(), "run_scores", running_scores.cpu(), "flags", flags.cpu(), "src", src_row.cpu()[:4, :4]); sys.stdout.flush()
O.beam_step = dbg_bs
out = model.generate(px.numpy(), num_beams=4, max_length=10, forced_bos_token_id=rc.vocab_size - 7)
print(out.sequences.cpu(), out.scores.cpu(), out["steps"])
