"""BASELINE configs[4]: the fp8 MFMA path (OCP e4m3 activations / weights, e5m2 gradients, per-tensor scaling — from the
tensor's current amax, or delayed: from the amax of the previous pass —, fp32 accumulate) for the QKV / FFN projections.  The reference has no fp8 mode (main.py:96-101 offers fp32 / fp16 / bf16), so the
path is judged the way the verdict asks: kernels against an exact dequantised reference, the model against the fp32 oracle with a
stated tolerance.

Tolerances: fp8 GEMM vs fp32 matmul of the SAME quantised operands: 1e-4 of the output scale with fp32 output (accumulation
order only; measured 2.2e-5 at K = 1024), bf16 rounding with bf16 output.  Model level (fp8 projections inside the bf16 model) vs the fp32 oracle: loss within
3e-2 relative, every large gradient leaf cosine > 0.9, all leaves together cosine > 0.97."""
import numpy as np
import pytest
import torch

from util_small import batch, make_pair

pytestmark = pytest.mark.gpu
FMAX = {torch.float8_e4m3fn: 448.0, torch.float8_e5m2: 57344.0}


def _quant_ref(x_bf16, fmt):
    """per-tensor current scaling exactly as csrc/fp8.hip does it, on the CPU"""
    x = x_bf16.float()
    amax = x.abs().max()
    scale = (torch.tensor(FMAX[fmt]) / amax) if amax > 0 else torch.tensor(1.0)
    q = (x * scale).clamp(-FMAX[fmt], FMAX[fmt]).to(fmt)
    return q, float(amax), float(amax / FMAX[fmt]) if amax > 0 else 1.0


@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("rows,cols", [(200, 136), (64, 64), (50, 768), (4096, 1024), (3, 8)])
def test_quantize_matches_reference_bytes(dev, fmt, rows, cols):
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) * torch.rand(rows, 1, generator=g) * 3).to(torch.bfloat16)
    x[rows // 2, cols // 3] = 37.5  # a clear maximum
    rp = (rows + 127) // 128 * 128
    q = torch.zeros((rows, cols), dtype=fmt, device=dev)
    qT = torch.full((cols, rp), 1.0, dtype=torch.float32, device=dev).to(fmt)  # pre-filled: the pad columns must be zeroed
    st = torch.zeros(2, device=dev)
    xd = x.to(dev)
    ops.fp8_quantize([ops.fp8_item(xd, rows, cols, st, fmt, q=q, qT=qT, rows_pad=rp)])
    torch.cuda.synchronize()
    ref, amax, sinv = _quant_ref(x, fmt)
    assert abs(st[0].item() - amax) == 0 and abs(st[1].item() - sinv) < 1e-7 * max(sinv, 1)
    got = q.cpu().view(torch.uint8)
    exp = ref.view(torch.uint8)
    # -0.0 vs +0.0 are both zero: compare values, then bytes away from zero
    assert torch.equal(q.cpu().float(), ref.float())
    assert (got[ref.float() != 0] == exp[ref.float() != 0]).all()
    gT = qT.cpu().float()
    assert torch.equal(gT[:, :rows], ref.float().T) and (gT[:, rows:] == 0).all()


def test_quantize_all_zero_tensor_and_grouping(dev):
    from mic_amd import ops

    z = torch.zeros((64, 64), dtype=torch.bfloat16, device=dev)
    q = torch.ones((64, 64), dtype=torch.float32, device=dev).to(torch.float8_e4m3fn)
    st = torch.zeros((12, 2), device=dev)
    items = [ops.fp8_item(z, 64, 64, st[0], torch.float8_e4m3fn, q=q)]
    xs, xd, qs = [], [], []
    g = torch.Generator().manual_seed(0)
    for i in range(1, 12):  # more than 8 items: two table launches
        x = (torch.randn(40 + 8 * i, 72, generator=g) * (i + 1)).to(torch.bfloat16)
        xs.append(x)
        xd.append(x.to(dev))  # the item structs hold raw device pointers: keep the tensors alive
        qs.append(torch.empty((x.shape[0], 72), dtype=torch.float8_e4m3fn, device=dev))
        items.append(ops.fp8_item(xd[-1], x.shape[0], 72, st[i], torch.float8_e4m3fn, q=qs[-1]))
    ops.fp8_quantize(items)
    torch.cuda.synchronize()
    assert st[0, 0].item() == 0 and st[0, 1].item() == 1.0 and (q.cpu().float() == 0).all()
    for i, (x, qq) in enumerate(zip(xs, qs), start=1):
        ref, amax, sinv = _quant_ref(x, torch.float8_e4m3fn)
        assert st[i, 0].item() == amax and torch.equal(qq.cpu().float(), ref.float()), i


@pytest.mark.parametrize("afmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("M,N,K,cdt", [(200, 136, 256, torch.float32), (37, 72, 128, torch.bfloat16), (1024, 1024, 1024, torch.float32),
                                       (4096, 1024, 1024, torch.bfloat16), (4096, 4096, 1024, torch.bfloat16), (3200, 2304, 768, torch.float32)])
def test_fp8_gemm_equals_dequantised_matmul(dev, afmt, M, N, K, cdt):
    """C = (qa sa)(qb sb)^T + bias with every tile configuration (64x64, 128x128 with and without K-groups, 256x256):
    exact up to fp32 accumulation order against an fp32 matmul of the dequantised operands."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    qa, _, sa = _quant_ref(a, afmt)
    qb, _, sb = _quant_ref(b, torch.float8_e4m3fn)
    sa_d, sb_d = torch.tensor([sa], device=dev), torch.tensor([sb], device=dev)
    out = torch.empty((M, N), dtype=cdt, device=dev)
    ops.gemm(qa.to(dev), qb.to(dev), out, M, N, K, bias=bias.to(dev), a_scale_inv=sa_d, b_scale_inv=sb_d)
    torch.cuda.synchronize()
    ref = (qa.float() @ qb.float().T) * (sa * sb) + bias
    err = ((out.float().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < (1e-4 if cdt == torch.float32 else 8e-3), err
    # and the quantisation error itself stays in the fp8 range against the unquantised product (sanity of the scaling)
    full = a.float() @ b.float().T + bias
    rel = ((ref - full).norm() / full.norm()).item()
    assert rel < (0.05 if afmt == torch.float8_e4m3fn else 0.09), rel


@pytest.mark.parametrize("afmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("M,N,K,kv", [(1024, 1024, 2432, 2404), (3072, 1024, 4096, 0), (128, 256, 128, 100), (4096, 1024, 2560, 2433),
                                      (768, 3072, 3200, 0), (144, 272, 384, 300)])
def test_fp8_gemm_kmajor_weight_gradient_form(dev, afmt, M, N, K, kv):
    """dW[M][N] = sum over rows k < k_valid of dy[k][M] x[k][N]: both fp8 operands K-MAJOR (as their producers wrote them, no
    transposed copies; fragments through ds_read_b64_tr_b8), fp32 output, rows at and behind k_valid count as zero whatever they
    hold (packed batches leave stale rows there).  128 x 128 tiles with one / two K-groups and 256 x 256 tiles; exact up to fp32
    accumulation order against an fp32 matmul of the dequantised operands; grouped like a layer's weight gradients."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    dy = (torch.randn(K, M, generator=g) * 0.3).to(torch.bfloat16)
    x = (torch.randn(K, N, generator=g) * 2).to(torch.bfloat16)
    qa, _, sa = _quant_ref(dy, afmt)
    qb, _, sb = _quant_ref(x, torch.float8_e4m3fn)
    valid = kv if kv else K
    ra, rb = qa.float().clone(), qb.float().clone()
    ra[valid:], rb[valid:] = 0, 0
    ref = (ra.T @ rb) * (sa * sb)
    sa_d, sb_d = torch.tensor([sa], device=dev), torch.tensor([sb], device=dev)
    out = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    ops.gemm(qa.to(dev), qb.to(dev), out, M, N, K, a_kmajor=True, b_kmajor=True, a_scale_inv=sa_d, b_scale_inv=sb_d, k_valid=kv)
    torch.cuda.synchronize()
    err = ((out.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-4, err
    # grouped with a second problem of another shape (one launch, the planner sees both)
    M2, N2 = 256, 1024
    dy2 = (torch.randn(K, M2, generator=g) * 0.1).to(torch.bfloat16)
    qa2, _, sa2 = _quant_ref(dy2, afmt)
    ra2 = qa2.float().clone()
    ra2[valid:] = 0
    N2 = min(N2, N)
    ref2 = (ra2.T @ rb[:, :N2]) * (sa2 * sb)
    sa2_d = torch.tensor([sa2], device=dev)
    out1 = torch.zeros((M, N), dtype=torch.float32, device=dev)
    out2 = torch.zeros((M2, N2), dtype=torch.float32, device=dev)
    qa_d, qb_d, qa2_d = qa.to(dev), qb.to(dev), qa2.to(dev)
    ops.gemm_grouped([ops.gemm_args(qa_d, qb_d, out1, M, N, K, a_kmajor=True, b_kmajor=True, a_scale_inv=sa_d, b_scale_inv=sb_d, k_valid=kv),
                      ops.gemm_args(qa2_d, qb_d, out2, M2, N2, K, a_kmajor=True, b_kmajor=True, a_scale_inv=sa2_d, b_scale_inv=sb_d, k_valid=kv)])
    torch.cuda.synchronize()
    assert ((out1.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-4
    assert ((out2.cpu() - ref2).abs().max() / ref2.abs().max()).item() < 1e-4


def test_fp8_gemm_split_k_slabs(dev):
    """the hoisted cross-k/v dX: [3200][1024] = dkv [3200][24576] . W over K = 24576 — too few output tiles to fill the chip, so the
    reduction is split into fp32 slabs (summed by mic_sum_slabs), as in bf16"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(11)
    M, N, K, nsp = 600, 256, 4096, 8
    a = (torch.randn(M, K, generator=g) * 0.2).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.3).to(torch.bfloat16)
    qa, _, sa = _quant_ref(a, torch.float8_e5m2)
    qb, _, sb = _quant_ref(b, torch.float8_e4m3fn)
    sa_d, sb_d = torch.tensor([sa], device=dev), torch.tensor([sb], device=dev)
    Mp = 640
    d32 = torch.full((nsp * Mp, N), 3.0, dtype=torch.float32, device=dev)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(qa.to(dev), qb.to(dev), d32, M, N, K, split_k=nsp, split_stride=Mp * N, a_scale_inv=sa_d, b_scale_inv=sb_d)
    ops.sum_slabs(d32, nsp, Mp * N, out, M, N, d32.stride(0), out.stride(0))
    torch.cuda.synchronize()
    ref = (qa.float() @ qb.float().T) * (sa * sb)
    assert ((out.float().cpu() - ref).abs().max() / ref.abs().max()).item() < 8e-3


def test_fp8_gemm_epilogue_grouped_and_errors(dev):
    from mic_amd import _lib as L
    from mic_amd import ops

    g = torch.Generator().manual_seed(5)
    M, N, K = 256, 384, 512
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16)
    z_in = torch.randn(M, N, generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16)
    qa, _, sa = _quant_ref(a, torch.float8_e5m2)
    qb, _, sb = _quant_ref(b, torch.float8_e4m3fn)
    sa_d, sb_d = torch.tensor([sa], device=dev), torch.tensor([sb], device=dev)
    # dX-style epilogue: v = acc * act'(Zin) + R, accumulate into C
    out = torch.ones((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(qa.to(dev), qb.to(dev), out, M, N, K, zin=z_in.to(dev), dact=L.ACT_GELU_TANH, residual=res.to(dev), accumulate=True,
             a_scale_inv=sa_d, b_scale_inv=sb_d)
    zf = z_in.float().requires_grad_(True)
    torch.nn.functional.gelu(zf, approximate="tanh").sum().backward()
    ref = (qa.float() @ qb.float().T) * (sa * sb) * zf.grad + res.float() + 1.0
    assert ((out.float().cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-2
    # grouped launch: three weight-gradient-like problems (fp32 outputs), one launch
    outs, refs, args, keep = [], [], [], []
    for i, (m, n, k) in enumerate([(1024, 256, 384), (256, 256, 128), (768, 512, 256)]):
        x = torch.randn(m, k, generator=g).to(torch.bfloat16)
        y = torch.randn(n, k, generator=g).to(torch.bfloat16)
        qx, _, sx = _quant_ref(x, torch.float8_e5m2)
        qy, _, sy = _quant_ref(y, torch.float8_e4m3fn)
        o = torch.empty((m, n), dtype=torch.float32, device=dev)
        dx_, dy_, sx_, sy_ = qx.to(dev), qy.to(dev), torch.tensor([sx], device=dev), torch.tensor([sy], device=dev)
        keep.append((dx_, dy_, sx_, sy_))  # the argument structs hold raw device pointers
        outs.append(o)
        refs.append((qx.float() @ qy.float().T) * (sx * sy))
        args.append(ops.gemm_args(dx_, dy_, o, m, n, k, a_scale_inv=sx_, b_scale_inv=sy_))
    ops.gemm_grouped(args)
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert ((o.cpu() - r).abs().max() / r.abs().max()).item() < 1e-4
    with pytest.raises(L.MicError, match="multiple of 128"):
        ops.gemm(qa.to(dev)[:, :64], qb.to(dev)[:, :64], out, M, N, 64, a_scale_inv=sa_d, b_scale_inv=sb_d)
    with pytest.raises(L.MicError, match="both k-major"):  # NN / mixed layouts do not exist in fp8: NT or TN
        ops.gemm(qa.to(dev), qb.to(dev), out, M, N, K, b_kmajor=True, a_scale_inv=sa_d, b_scale_inv=sb_d)


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.reshape(-1).double(), b.reshape(-1).double(), dim=0).item()


@pytest.mark.parametrize("head,ls", [("0", 0.0), ("bwd", 0.0), ("all", 0.0), ("all", 0.1)])
@pytest.mark.parametrize("compact", [False, True])
def test_fp8_train_step_against_fp32_oracle(dev, compact, head, ls, monkeypatch):
    """head: MIC_FP8_HEAD — the tied LM head's GEMMs in the storage dtype / its two backward GEMMs on fp8 operands (default) / the
    forward projection too"""
    from mic_amd import loss_rows
    from oracle import train_ref

    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0, d_model=256, d_ffn=512, d_heads=4,
                             v_hidden=256, v_ffn=512, v_heads=4)
    monkeypatch.setenv("MIC_FP8_HEAD", head)
    model.engine.set_gemm_dtype("fp8")
    assert model.engine.fp8_head == {"0": 0, "bwd": 1, "all": 2}[head] and ("shared" in model.engine._w8) == (head != "0")
    B, T = 4, 16
    px, labels, mask, dec_in = batch(rc, B, T, seed=9)
    ref_loss, ref_g = train_ref.loss_and_grads(rc, p, px, labels, mask, dec_in, label_smoothing_factor=ls)
    d = model._dev
    pos = torch.arange(T, dtype=torch.int32, device=dev)[None].expand(B, T).contiguous()
    kw = {}
    if compact:
        idx, rl = loss_rows(mask.numpy(), labels.numpy())
        kw = dict(rows=(d(idx, torch.int32), len(idx)), row_labels=d(rl, torch.int32))
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T, label_smoothing=ls, **kw)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) < 3e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    got = model.store.export_flat("grad")
    worst = {}
    for k, rg in ref_g.items():
        if rg.abs().max().item() > 1e-4 and rg.numel() >= 4096:
            worst[k] = _cos(torch.from_numpy(got[k]).reshape(rg.shape), rg)
    bad = {k: v for k, v in worst.items() if v < 0.9}
    assert not bad, sorted(bad.items(), key=lambda kv: kv[1])[:6]
    a = torch.cat([torch.from_numpy(got[k]).reshape(-1) for k in ref_g])
    b = torch.cat([v.reshape(-1) for v in ref_g.values()])
    assert _cos(a, b) > 0.97, _cos(a, b)
    # the fp8 projections really ran in fp8: their quantised weights exist and carry the weights' scale
    w8 = model.engine._w8["dec0.fc1"]
    wref, amax, sinv = _quant_ref(model.store.w("dec0.fc1.w").cpu(), torch.float8_e4m3fn)
    assert torch.equal(w8[0].cpu().float(), wref.float()) and torch.equal(w8[1].cpu().float(), wref.float().T) and abs(w8[2][1].item() - sinv) < 1e-9
    if head != "0":  # ... and so did the tied head: e4m3 copies of E and E^T under the embedding's scale
        e8 = model.engine._w8["shared"]
        eref, _, esinv = _quant_ref(model.store.w("shared").cpu(), torch.float8_e4m3fn)
        assert torch.equal(e8[0].cpu().float(), eref.float()) and torch.equal(e8[1].cpu().float(), eref.float().T) and abs(e8[2][1].item() - esinv) < 1e-9


def test_fp8_head_switch_rejects_unknown_values(dev, monkeypatch):
    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0, d_model=256, d_ffn=512, d_heads=4,
                             v_hidden=256, v_ffn=512, v_heads=4)
    monkeypatch.setenv("MIC_FP8_HEAD", "fwd")
    with pytest.raises(ValueError, match="MIC_FP8_HEAD"):
        model.engine.set_gemm_dtype("fp8")
    model.engine.set_gemm_dtype("fp8", head="bwd")  # the argument goes before the environment
    assert model.engine.fp8_head == 1
    from mic_amd import Trainer, create_learning_rate_fn

    Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 1e-3), gemm_dtype="fp8", fp8_head="0")
    assert model.engine.fp8_head == 0 and "shared" not in model.engine._w8
    with pytest.raises(ValueError, match="head"):
        Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 1e-3), gemm_dtype="fp8", fp8_head="forward")


def test_fp8_trainer_learns_and_eval_matches(dev):
    from mic_amd import Trainer, create_learning_rate_fn

    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.1)
    tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 2e-3), gemm_dtype="fp8")
    assert model.engine.fp8
    px, labels, mask, dec_in = batch(rc, 2, 12, seed=4)
    b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
    l0 = float(tr.eval_step(b)["loss"])
    for _ in range(6):
        out = tr.train_step(b)
    l1 = float(tr.eval_step(b)["loss"])
    assert np.isfinite(float(out["loss"])) and l1 < l0 - 0.5, (l0, l1)  # weights are re-quantised after every optimizer step
    # generation keeps working next to the fp8 trainer (decode runs in the storage dtype)
    seq = model.generate(px.numpy(), num_beams=2, max_length=6).sequences
    assert tuple(seq.shape) == (2, 6)
    with pytest.raises(ValueError, match="bfloat16"):
        _, _, m32 = make_pair(torch.float32, dev)
        m32.engine.set_gemm_dtype("fp8")


@pytest.mark.parametrize("overlap", [True, False])
def test_fp8_weights_requantised_behind_the_optimizer_give_the_same_steps(dev, overlap):
    """opt-in placement of the per-step weight quantisation (MIC_FP8_REQUANT_OPT=1: per gradient bucket behind its AdamW pass, on
    the optimizer's stream) against the default (all weights at the start of the next pass): same scales, same bytes, same losses;
    and both equal a third trainer whose fused emission is off (every operand through mic_fp8_quantize)"""
    from mic_amd import Trainer, create_learning_rate_fn

    losses = []
    for mode in ("default", "requant_opt", "unfused"):
        rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.1)
        tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 2e-3), gemm_dtype="fp8", overlap_optimizer=overlap, bucket_mb=0.25)
        eng = model.engine
        eng._w8_requant_opt = mode == "requant_opt"
        if mode == "unfused":
            eng.fp8_fused = False
        px, labels, mask, dec_in = batch(rc, 3, 12, seed=9)
        b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
        ls = [float(tr.train_step(b)["loss"]) for _ in range(5)]
        assert not eng._w8_stale and eng._w8_event is not None  # the next pass finds its fp8 weight copies made (it waits for the event)
        losses.append(ls)
    assert losses[0] == losses[1], losses
    # fused and unfused emission write the same bytes; what differs is the summation order of fp32 atomics downstream
    assert max(abs(a - b) for a, b in zip(losses[0], losses[2])) < 2e-4 * abs(losses[0][0]), losses


def test_fp8_weight_copies_made_on_a_lagging_side_stream_are_waited_for(dev):
    """The next step's fp8 weight copies (and the two e4m3 copies of the tied embedding the fp8 head reads) are made on the aux stream
    behind the optimizer; every fp8 GEMM that reads one waits for that stream's event.  Here the aux stream is made slow (a 2-ms spin
    kernel in front of each refresh): the losses must equal those of a trainer that re-quantises on the step's own stream."""
    from mic_amd import Trainer, create_learning_rate_fn, ops

    losses = []
    for lag in (True, False):
        rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.1, d_model=256, d_ffn=512, d_heads=4,
                                 v_hidden=256, v_ffn=512, v_heads=4)
        tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 2e-3), gemm_dtype="fp8")
        eng = model.engine
        assert eng.fp8_head == 2 and "shared" in eng._w8
        if lag:
            a, b = torch.zeros(1 << 18, device=dev), torch.zeros(1 << 18, device=dev)
            refresh = eng.fp8_refresh_weights

            def lagging(side, upto=None, wait_event=None):
                with torch.cuda.stream(side):
                    with ops.pinned_stream():
                        ops.comm_emulate(a, b, 1 << 20, 2000.0, 8)
                return refresh(side, upto=upto, wait_event=wait_event)
            eng.fp8_refresh_weights = lagging
        else:
            eng._w8_async = False
        px, labels, mask, dec_in = batch(rc, 3, 12, seed=9)
        bt = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
        losses.append([float(tr.train_step(bt)["loss"]) for _ in range(5)])
    assert max(abs(x - y) for x, y in zip(*losses)) < 2e-4 * abs(losses[1][0]), losses


def test_fp8_trainer_inference_entry_points_stay_in_the_storage_dtype(dev):
    """`encode` / `decode` / `generate` beside an fp8 trainer run bf16 GEMMs: identical to a bf16 engine holding the same weights,
    (i) directly after Trainer construction, when no pass has quantised the fp8 weight copies yet, and (ii) directly after a
    train_step, when those copies are one optimizer step stale."""
    from mic_amd import Trainer, create_learning_rate_fn

    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    _, _, plain = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
    tr = Trainer(model, create_learning_rate_fn(64, 2, 4, 2, 1e-3), gemm_dtype="fp8")
    tp = Trainer(plain, create_learning_rate_fn(64, 2, 4, 2, 1e-3))
    assert model.engine.fp8 and not plain.engine.fp8
    px, labels, mask, dec_in = batch(rc, 2, 12, seed=4)

    def check(tag):
        a, b = model.encode(px.numpy()), plain.encode(px.numpy())
        assert torch.equal(a.last_hidden_state, b.last_hidden_state), tag
        assert torch.isfinite(a.last_hidden_state.float()).all()
        la = model.decode(dec_in.numpy(), a, decoder_attention_mask=mask.numpy()).logits
        lb = plain.decode(dec_in.numpy(), b, decoder_attention_mask=mask.numpy()).logits
        assert torch.equal(la, lb), tag
        sa = model.generate(px.numpy(), num_beams=2, max_length=6)
        sb = plain.generate(px.numpy(), num_beams=2, max_length=6)
        assert torch.equal(sa.sequences, sb.sequences) and torch.equal(sa.scores, sb.scores), tag
        assert model.engine.fp8  # the switch is restored behind every entry point

    check("after construction")
    b = {"pixel_values": px.numpy(), "input_ids": labels.numpy(), "attention_mask": mask.numpy(), "decoder_input_ids": dec_in.numpy()}
    tr.train_step(b)
    # give the bf16 model the fp8 trainer's updated weights (its own step would differ by the fp8 gradients)
    plain.params = model.params
    check("after a train step")


# ---------------------------------------------------------------------------------------------- delayed scaling
@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("rows,cols", [(200, 136), (4096, 1024), (3, 8)])
def test_quantize_delayed_uses_given_scale_records_amax_and_saturates(dev, fmt, rows, cols):
    """mic_fp8_quantize alone (no amax pass): scale from state[0] as given, max |x| of the pass into amax_next, values beyond
    the old amax clamp to +-FMAX; mic_fp8_roll_amax then makes the recorded amax current."""
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2).to(torch.bfloat16)
    x[0, 0] = 50.0   # beyond the "old" amax below -> saturates
    x[rows - 1, cols - 1] = -64.0
    old_amax = 16.0
    rp = (rows + 127) // 128 * 128
    P = ops.fp8_amax_partials()
    st = torch.tensor([[old_amax, 0.0], [3.0, 0.0]], device=dev)   # slot 1: a neighbour with one recorded partial
    part = torch.zeros((2, P), device=dev)
    part[1, 5] = 7.0
    q = torch.zeros((rows, cols), dtype=fmt, device=dev)
    qT = torch.zeros((cols, rp), dtype=fmt, device=dev)
    xd = x.to(dev)
    ops.fp8_quantize([ops.fp8_item(xd, rows, cols, st[0], fmt, q=q, qT=qT, rows_pad=rp, amax_next=part[0])], amax_pass=False)
    torch.cuda.synchronize()
    fm = FMAX[fmt]
    ref = (x.float() * (fm / old_amax)).clamp(-fm, fm).to(fmt)
    assert torch.equal(q.cpu().float(), ref.float()) and torch.equal(qT.cpu().float()[:, :rows], ref.float().T)
    assert q.cpu().float()[0, 0].item() == fm and q.cpu().float()[rows - 1, cols - 1].item() == -fm
    assert st[0, 0].item() == old_amax and abs(st[0, 1].item() - old_amax / fm) < 1e-7 * old_amax / fm
    tiles = (rp // 128) * ((cols + 63) // 64)   # the quantiser walks 128 x 64 tiles, one partial per tile
    assert part[0].max().item() == 64.0 and int((part[0] > 0).sum()) == min(tiles, P) and part[1].max().item() == 7.0
    ops.fp8_roll_amax(st, part, 2)
    torch.cuda.synchronize()
    assert st[0, 0].item() == 64.0 and st[1, 0].item() == 7.0 and float(part.abs().max()) == 0.0
    ops.fp8_roll_amax(st, part, 2)   # nothing recorded since: the amax in use stays
    torch.cuda.synchronize()
    assert st[0, 0].item() == 64.0 and st[1, 0].item() == 7.0


def _same(ga, gb, tol=1e-5):
    """leaf-wise equality up to the summation order of the fp32 atomics some gradient kernels use"""
    for k in ga:
        a, b = ga[k].astype(np.float64), gb[k].astype(np.float64)
        assert np.abs(a - b).max() <= tol * max(np.abs(a).max(), 1e-30), (k, np.abs(a - b).max(), np.abs(a).max())


def _fp8_pass(model, rc, px, labels, mask, dec_in, B, T):
    d = model._dev
    dev = model.device
    pos = torch.arange(T, dtype=torch.int32, device=dev)[None].expand(B, T).contiguous()
    loss = model.engine.loss_and_grads(d(px, torch.float32), d(dec_in, torch.int32).reshape(-1), pos.reshape(-1), d(mask, torch.int32),
                                       d(labels, torch.int32).reshape(-1), B, T)
    torch.cuda.synchronize()
    return loss.item(), {k: v.copy() for k, v in model.store.export_flat("grad").items()}


@pytest.mark.parametrize("T", [12, 72])
def test_fp8_fused_emission_equals_the_quantiser_path_also_beyond_one_attention_tile(dev, T):
    """engine-level: three passes on one batch with the producers emitting the fp8 operands (from pass 2 on) against the same three
    passes with every operand through mic_fp8_quantize (`fp8_fused = False`): same losses, same gradients up to the fp32 atomics'
    order.  T = 72: more than one 64-key attention tile — the tiled attention backward has no fp8 form, its dQ / dK / dV go through the
    quantiser while LayerNorm / GELU / dGELU still emit."""
    res = {}
    for fused in (True, False):
        rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0)
        model.engine.set_gemm_dtype("fp8")
        model.engine.fp8_fused = fused
        px, labels, mask, dec_in = batch(rc, 3, T, seed=21)
        out = [_fp8_pass(model, rc, px, labels, mask, dec_in, 3, T) for _ in range(3)]
        res[fused] = out
        if fused:
            assert len(model.engine._a8_ready) > 20
    for (la, ga), (lb, gb) in zip(res[True], res[False]):
        assert abs(la - lb) <= 1e-5 * abs(lb), (la, lb)
        _same(ga, gb, tol=2e-4)


def test_fp8_delayed_scaling_reproduces_current_scaling_on_a_repeated_batch(dev):
    """Pass 1 of a fresh engine has no history: every tensor is scaled by its current amax.  Pass 2 on the SAME batch and weights
    runs delayed (one quantiser pass per tensor, scale = amax recorded in pass 1 = the current amax): loss and gradients must
    come out identical — the delayed path differs from the parity-checked current path only in where the amax comes from.
    A third pass on a 4x brighter image saturates some activations for one pass and must stay finite and close."""
    rc, p, model = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0, d_model=256, d_ffn=512, d_heads=4,
                             v_hidden=256, v_ffn=512, v_heads=4)
    model.engine.set_gemm_dtype("fp8", scaling="delayed")
    B, T = 4, 16
    px, labels, mask, dec_in = batch(rc, B, T, seed=9)
    eng = model.engine
    l1, g1 = _fp8_pass(model, rc, px, labels, mask, dec_in, B, T)
    assert not eng._a8_ready                      # pass 1: nothing had history
    n_slots = len(eng._a8_slots)
    assert n_slots > 0 and (eng._a8_part[:n_slots].max(dim=1).values > 0).all()   # every slot recorded its amax for the next pass
    model.store.grad.zero_()
    l2, g2 = _fp8_pass(model, rc, px, labels, mask, dec_in, B, T)
    assert eng._a8_ready == set(eng._a8_slots)    # pass 2: every tensor took the delayed path
    assert abs(l1 - l2) <= 1e-6 * abs(l1)
    _same(g1, g2)
    # current-scaling engine on the same batch: same numbers (it is the same arithmetic with the amax taken in-pass)
    rc2, p2, model2 = make_pair(torch.bfloat16, dev, gelu="tanh", decoder_ln_eps=1e-6, dropout=0.0, d_model=256, d_ffn=512, d_heads=4,
                                v_hidden=256, v_ffn=512, v_heads=4)
    model2.engine.set_gemm_dtype("fp8", scaling="current")
    l3, g3 = _fp8_pass(model2, rc2, px, labels, mask, dec_in, B, T)
    assert abs(l3 - l1) <= 1e-6 * abs(l1)
    _same(g1, g3)
    # outgrowing the recorded amax: saturation for one pass, finite and close to the current-scaling result
    model.store.grad.zero_()
    model2.store.grad.zero_()
    l4, g4 = _fp8_pass(model, rc, px * 4, labels, mask, dec_in, B, T)
    l5, g5 = _fp8_pass(model2, rc2, px * 4, labels, mask, dec_in, B, T)
    assert np.isfinite(l4) and abs(l4 - l5) < 0.1 * abs(l5), (l4, l5)
    a = np.concatenate([v.reshape(-1) for v in g4.values()])
    b = np.concatenate([v.reshape(-1) for v in g5.values()])
    assert np.isfinite(a).all() and _cos(torch.from_numpy(a), torch.from_numpy(b)) > 0.8
    # the pass after that has (nearly) caught up: its scales come from the bright pass, whose downstream amaxes were themselves
    # taken behind saturated inputs, so it is close to — not bit-equal with — current scaling
    model.store.grad.zero_()
    model2.store.grad.zero_()
    l6, g6 = _fp8_pass(model, rc, px * 4, labels, mask, dec_in, B, T)
    l7, g7 = _fp8_pass(model2, rc2, px * 4, labels, mask, dec_in, B, T)
    assert abs(l6 - l7) <= 2e-3 * abs(l7), (l6, l7)
    a = np.concatenate([v.reshape(-1) for v in g6.values()])
    b = np.concatenate([v.reshape(-1) for v in g7.values()])
    assert _cos(torch.from_numpy(a), torch.from_numpy(b)) > 0.995
