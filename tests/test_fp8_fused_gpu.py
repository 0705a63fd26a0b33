"""BASELINE configs[4], fused fp8 emission: the producers of the fp8 GEMM operands (LayerNorm forward / backward, the GELU / dGELU
GEMM epilogues, attention backward) write the e4m3 / e5m2 bytes themselves under the tensor's DELAYED scale instead of leaving a
bf16 tensor for a quantiser launch (181 launches per train step).  The reference has no fp8 mode (main.py:96-101), so the fused
kernels are judged against the parity-checked two-kernel path of this build: SAME bytes (bit-exact, both zeros count as equal),
same dequantisation factor, same recorded amax — on scales that make part of the tensor saturate."""
import pytest
import torch

pytestmark = pytest.mark.gpu
FMAX = {torch.float8_e4m3fn: 448.0, torch.float8_e5m2: 57344.0}
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2


def _two_kernel(x_bf16, rows, cols, fmt, old_amax, dev):
    """the unfused path: mic_fp8_quantize of the bf16 tensor under delayed scaling -> (bytes as float, 1/scale, recorded amax)"""
    from mic_amd import ops

    st = torch.tensor([old_amax, 0.0], device=dev)
    part = torch.zeros(ops.fp8_amax_partials(), device=dev)
    q = torch.zeros((rows, cols), dtype=fmt, device=dev)
    ops.fp8_quantize([ops.fp8_item(x_bf16, rows, cols, st, fmt, q=q, amax_next=part)], amax_pass=False)
    torch.cuda.synchronize()
    return q.float().cpu(), st[1].item(), part.max().item()


def _fused_slot(old_amax, dev):
    from mic_amd import ops

    return torch.tensor([old_amax, 0.0], device=dev), torch.zeros(ops.fp8_amax_partials(), device=dev)


@pytest.mark.parametrize("rows,width,p", [(2404, 1024, 0.0), (3200, 768, 0.0), (37, 256, 0.1), (130, 2048, 0.0)])
def test_layernorm_fwd_q8_equals_layernorm_then_quantize(dev, rows, width, p):
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + width)
    x = (torch.randn(rows, width, generator=g) * 3 + 0.5).to(torch.bfloat16).to(dev)
    gam = (torch.rand(width, generator=g) + 0.5).to(dev)
    bet = (torch.randn(width, generator=g) * 0.2).to(dev)
    y = torch.zeros_like(x)
    mean, rstd = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    ops.layernorm_fwd(x, gam, bet, 1e-5, y, mean, rstd, rows=rows, dropout_p=p, dropout_seed=77)
    old = 2.5  # below the tensor's maximum: the tail saturates
    ref, sinv, amax = _two_kernel(y, rows, width, E4, old, dev)
    for keep_y in (True, False):
        st, part = _fused_slot(old, dev)
        q = torch.zeros((rows + 3, width), dtype=E4, device=dev)
        y2 = torch.zeros_like(x) if keep_y else None
        m2, r2 = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
        ops.layernorm_fwd(x, gam, bet, 1e-5, y2, m2, r2, rows=rows, dropout_p=p, dropout_seed=77, q8=ops.fp8_out(q, st, part))
        torch.cuda.synchronize()
        assert torch.equal(q[:rows].float().cpu(), ref) and float(q[rows:].float().abs().max()) == 0.0
        assert st[1].item() == sinv and part.max().item() == amax and torch.equal(m2, mean) and torch.equal(r2, rstd)
        if keep_y:
            assert torch.equal(y2, y)
    assert (ref.abs() == 448.0).any() and amax > old


@pytest.mark.parametrize("rows,width,p,of_dx", [(2404, 1024, 0.1, False), (3200, 768, 0.0, True), (100, 2048, 0.1, False), (100, 2048, 0.0, True),
                                                (64, 256, 0.0, False)])
def test_layernorm_bwd_q8_equals_backward_then_quantize(dev, rows, width, p, of_dx):
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows * 3 + width)
    x = torch.randn(rows, width, generator=g).to(torch.bfloat16).to(dev)
    dy = (torch.randn(rows, width, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    dres = (torch.randn(rows, width, generator=g) * 0.01).to(torch.bfloat16).to(dev)
    gam = (torch.rand(width, generator=g) + 0.5).to(dev)
    mean, rstd = x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
    nb = ops.layernorm_bwd_blocks(rows)
    dx, dxm = torch.zeros_like(x), torch.zeros_like(x)
    part = torch.zeros((2 * nb, width), device=dev)
    ops.layernorm_bwd_partials(x, gam, mean, rstd, dy, dx, part, rows=rows, dres=dres, dxm=None if of_dx else dxm, dropout_p=p, dropout_seed=5)
    src = dx if of_dx else dxm
    old = float(src.float().abs().max()) * 0.6
    ref, sinv, amax = _two_kernel(src, rows, width, E5, old, dev)
    for keep in ((True,) if of_dx else (True, False)):
        st, pt = _fused_slot(old, dev)
        q = torch.zeros((rows, width), dtype=E5, device=dev)
        dx2 = torch.zeros_like(x)
        dxm2 = torch.zeros_like(x) if (keep and not of_dx) else None
        part2 = torch.zeros_like(part)
        ops.layernorm_bwd_partials(x, gam, mean, rstd, dy, dx2, part2, rows=rows, dres=dres, dxm=dxm2, dropout_p=p, dropout_seed=5,
                                   q8=ops.fp8_out(q, st, pt), q8_of_dx=of_dx)
        torch.cuda.synchronize()
        assert torch.equal(dx2, dx) and torch.equal(part2, part)
        if dxm2 is not None:
            assert torch.equal(dxm2, dxm)
        assert torch.equal(q.float().cpu(), ref) and st[1].item() == sinv and pt.max().item() == amax
    assert (ref.abs() == 57344.0).any()


def _attn_inputs(B, H, Tq, Tk, rows_q, rows_k, g, dev, self_attn):
    d = H * 64
    if self_attn:
        qkv = (torch.randn(rows_q, 3 * d, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        return qkv, qkv[:, d:], qkv[:, 2 * d:], 3 * d, 3 * d
    q = (torch.randn(rows_q, d, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    kv = (torch.randn(rows_k, 2 * d, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    return q, kv, kv[:, d:], d, 2 * d


@pytest.mark.parametrize("mode", ["self_dense", "cross_dense", "self_packed", "cross_packed", "vit"])
def test_attention_bwd_q8_equals_backward_then_quantize(dev, mode):
    """dQ / dK / dV as e5m2 bytes: self-attention writes ONE [rows][3d] tensor (one scale), cross-attention dQ [rows][d] and
    dK | dV [encoder rows][2d] (two scales); dense rows, packed rows (ragged sequences), the ViT's 50 x 50 tile"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(len(mode))
    B, H = 5, 4
    d = H * 64
    self_attn = mode.startswith("self") or mode == "vit"
    packed = mode.endswith("packed")
    Tq = 50 if mode == "vit" else 64
    Tk = Tq if self_attn else 50
    if packed:
        q_len = torch.tensor([64, 9, 33, 1, 40], dtype=torch.int32)
        q_off = torch.cumsum(q_len, 0, dtype=torch.int32) - q_len
        rows_q = int(q_len.sum())
        q_off_d, q_len_d = q_off.to(dev), q_len.to(dev)
    else:
        rows_q, q_off_d, q_len_d = B * Tq, None, None
    rows_k = rows_q if self_attn else B * Tk
    q, k, v, ldq, ldk = _attn_inputs(B, H, Tq, Tk, rows_q, rows_k, g, dev, self_attn)
    ctx = torch.zeros((rows_q, d), dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B * H * Tq, device=dev)
    causal = mode.startswith("self")
    if packed:
        ops.attn_fwd_packed(q, k, v, ctx, B, H, Tq, Tk, q_off_d, q_len_d, kv_packed=self_attn, ldq=ldq, ldk=ldk, ldv=ldk, ldo=d, causal=causal, lse=lse)
    else:
        ops.attn_fwd(q, k, v, ctx, B, H, Tq, Tk, ldq=ldq, ldk=ldk, ldv=ldk, ldo=d, causal=causal, lse=lse)
    dctx = (torch.randn(rows_q, d, generator=g) * 0.02).to(torch.bfloat16).to(dev)
    # the bf16 reference gradients
    if self_attn:
        dqkv = torch.zeros((rows_q, 3 * d), dtype=torch.bfloat16, device=dev)
        dq, dk, dv, lddq, lddk = dqkv, dqkv[:, d:], dqkv[:, 2 * d:], 3 * d, 3 * d
    else:
        dq = torch.zeros((rows_q, d), dtype=torch.bfloat16, device=dev)
        dkv = torch.zeros((rows_k, 2 * d), dtype=torch.bfloat16, device=dev)
        dk, dv, lddq, lddk = dkv, dkv[:, d:], d, 2 * d
    if packed:
        ops.attn_bwd_packed(q, k, v, ctx, dctx, lse, dq, dk, dv, B, H, Tq, Tk, q_off_d, q_len_d, kv_packed=self_attn, ldq=ldq, ldk=ldk, ldv=ldk, ldo=d,
                            lddo=d, lddq=lddq, lddk=lddk, lddv=lddk, causal=causal)
    else:
        ops.attn_bwd(q, k, v, ctx, dctx, lse, dq, dk, dv, B, H, Tq, Tk, ldq=ldq, ldk=ldk, ldv=ldk, ldo=d, lddo=d, lddq=lddq, lddk=lddk, lddv=lddk,
                     causal=causal)
    torch.cuda.synchronize()
    if self_attn:
        old = float(dqkv.float().abs().max()) * 0.5
        ref, sinv, amax = _two_kernel(dqkv, rows_q, 3 * d, E5, old, dev)
        st, pt = _fused_slot(old, dev)
        q8 = torch.zeros((rows_q, 3 * d), dtype=E5, device=dev)
        ops.attn_bwd_q8(q, k, v, ctx, dctx, lse, ops.fp8_out(q8, st, pt), ops.fp8_out(q8[:, d:], st, pt), q8[:, 2 * d:], B, H, Tq, Tk, ldq=ldq, ldk=ldk,
                        ldv=ldk, ldo=d, lddo=d, q_off=q_off_d, q_len=q_len_d, kv_packed=packed, causal=causal)
        torch.cuda.synchronize()
        assert torch.equal(q8.float().cpu(), ref) and st[1].item() == sinv and pt.max().item() == amax
        assert (ref.abs() == 57344.0).any()
    else:
        oq, okv = float(dq.float().abs().max()) * 0.5, float(dkv.float().abs().max()) * 2.0
        refq, sq, aq = _two_kernel(dq, rows_q, d, E5, oq, dev)
        refkv, skv, akv = _two_kernel(dkv, rows_k, 2 * d, E5, okv, dev)
        stq, ptq = _fused_slot(oq, dev)
        stk, ptk = _fused_slot(okv, dev)
        q8 = torch.zeros((rows_q, d), dtype=E5, device=dev)
        kv8 = torch.zeros((rows_k, 2 * d), dtype=E5, device=dev)
        ops.attn_bwd_q8(q, k, v, ctx, dctx, lse, ops.fp8_out(q8, stq, ptq), ops.fp8_out(kv8, stk, ptk), kv8[:, d:], B, H, Tq, Tk, ldq=ldq, ldk=ldk,
                        ldv=ldk, ldo=d, lddo=d, q_off=q_off_d, q_len=q_len_d, kv_packed=False, causal=causal)
        torch.cuda.synchronize()
        assert torch.equal(q8.float().cpu(), refq) and stq[1].item() == sq and ptq.max().item() == aq
        assert torch.equal(kv8.float().cpu(), refkv) and stk[1].item() == skv and ptk.max().item() == akv


@pytest.mark.parametrize("M,N,K,kind", [(2404, 4096, 1024, "act"), (3200, 3072, 768, "act"), (2404, 4096, 1024, "dact"), (200, 256, 128, "act"),
                                        (4096, 4096, 1024, "dact"), (100, 136, 256, "dact")])
def test_fp8_gemm_fp8_output_equals_gemm_then_quantize(dev, M, N, K, kind):
    """the two GEMM epilogues whose result only ever feeds another fp8 GEMM: GELU(FFN-in) (pre-activation kept in bf16 for backward)
    and the dGELU-scaled dX of FFN-out; 64 / 128 tiles with and without K-groups (a 256-tile shape re-plans to 128)"""
    from mic_amd import _lib as L
    from mic_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    afmt = E4 if kind == "act" else E5
    a = (torch.randn(M, K, generator=g) * 2).to(afmt).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.5).to(E4).to(dev)
    sa, sb = torch.tensor([0.01], device=dev), torch.tensor([0.02], device=dev)
    bias = torch.randn(N, generator=g).to(dev)
    zin = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    z = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    kw = dict(bias=bias, act=L.ACT_GELU_TANH, zout=z) if kind == "act" else dict(zin=zin, dact=L.ACT_QUICK_GELU)
    ops.gemm(a, b, out, M, N, K, a_scale_inv=sa, b_scale_inv=sb, **kw)
    ofmt = E4 if kind == "act" else E5
    old = float(out.float().abs().max()) * 0.7
    ref, sinv, amax = _two_kernel(out, M, N, ofmt, old, dev)
    st, pt = _fused_slot(old, dev)
    q = torch.zeros((M, N), dtype=ofmt, device=dev)
    z2 = torch.zeros_like(z)
    kw2 = dict(bias=bias, act=L.ACT_GELU_TANH, zout=z2) if kind == "act" else dict(zin=zin, dact=L.ACT_QUICK_GELU)
    ops.gemm(a, b, q, M, N, K, a_scale_inv=sa, b_scale_inv=sb, c_q8=(st, pt), **kw2)
    torch.cuda.synchronize()
    assert torch.equal(q.float().cpu(), ref) and st[1].item() == sinv and pt.max().item() == amax
    if kind == "act":
        assert torch.equal(z2, z)
    assert (ref.abs() == FMAX[ofmt]).any()
    with pytest.raises(L.MicError, match="fp8 C"):  # the bare epilogue has no fp8 form
        ops.gemm(a, b, q, M, N, K, a_scale_inv=sa, b_scale_inv=sb, c_q8=(st, pt))


@pytest.mark.parametrize("fmt", [E4, E5])
def test_colsum_q8_grouped(dev, fmt):
    from mic_amd import ops

    g = torch.Generator().manual_seed(3)
    items, refs, keep = [], [], []
    for i, (rows, cols) in enumerate([(2404, 4096), (3200, 768), (7, 8), (130, 1032), (64, 256), (50, 3072), (999, 1024), (1, 16), (2404, 3072)]):
        x = (torch.randn(rows + 5, cols, generator=g) * (i + 1)).to(fmt).to(dev)  # rows behind `rows` must not count
        out = torch.full((cols,), 0.5, device=dev)
        sinv = torch.tensor([0.25 * (i + 1)], device=dev)
        keep.append((x, out, sinv))
        items.append((x, out, sinv, rows, cols))
        refs.append(0.5 + x[:rows].float().double().sum(0).cpu() * 0.25 * (i + 1))
    ops.colsum_q8_grouped(items)  # nine items: two launches
    torch.cuda.synchronize()
    for (x, out, sinv, rows, cols), r in zip(items, refs):
        assert ((out.double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-9)).item() < 1e-5, (rows, cols)


@pytest.mark.parametrize("rows,V,Vpad,ls", [(2404, 250054, 250112, 0.0), (100, 1000, 1024, 0.1), (64, 505, 512, 0.0), (3, 130, 136, 0.0)])
def test_ce_bwd_q8_against_closed_form_scale(dev, rows, V, Vpad, ls):
    """the fp8 LM head's CE backward: ONE e5m2 copy of dlogits, q = e5m2(mask (softmax - soft_label) * 57344) — the tensor's scale in
    closed form, no amax history — with the dequantisation factor 1 / (denom * 57344) beside it; the logits stay; the column sums of the
    gradient are added to the bias gradient; dequantised, it is mic_ce_bwd's in-place gradient to e5m2 precision"""
    from mic_amd import ops

    g = torch.Generator().manual_seed(rows + V)
    logits = ((torch.rand(rows, Vpad, generator=g) - 0.5) * 12).to(torch.bfloat16)
    logits[:, 7] += 9.0  # some rows are confident about column 7 ...
    labels = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    labels[::3] = 7      # ... and a third of the labels agree with them
    mask = (torch.rand(rows, generator=g) > 0.2).to(torch.int32)
    logits, labels, mask = logits.to(dev), labels.to(dev), mask.to(dev)
    lse = torch.logsumexp(logits[:, :V].float(), dim=1).contiguous()
    denom = mask.sum().float().reshape(1)
    st = torch.zeros(2, device=dev)
    q = torch.zeros((rows + 2, Vpad), dtype=E5, device=dev)
    keep = logits.clone()
    flb = torch.full((Vpad,), 0.25, device=dev)
    ops.ce_bwd_q8(logits, Vpad, V, Vpad, labels, mask, ls, lse, denom, rows, ops.fp8_out(q, st), colsum=flb)
    torch.cuda.synchronize()
    assert torch.equal(logits, keep)
    soft = torch.full((rows, V), ls / (V - 1) if ls > 0 else 0.0, device=dev)
    soft[torch.arange(rows, device=dev), labels.long()] = 1.0 - ls
    gref = (torch.softmax(logits[:, :V].float(), dim=1) - soft) * mask[:, None].float()
    want = (gref * 57344.0).to(E5).float()
    got = q[:rows, :V].float()
    # the same bytes up to the device's exp (a handful of entries on a rounding boundary land on the neighbouring e5m2 value)
    diff = (got != want)
    assert diff.float().mean().item() < 2e-4, diff.float().mean().item()
    assert float(((got - want).abs() / want.abs().clamp_min(2.0 ** -16))[diff].max() if diff.any() else 0.0) <= 0.34
    assert float(q[rows:].float().abs().max()) == 0.0 and float(q[:rows, V:].float().abs().max()) == 0.0
    assert abs(st[1].item() * denom.item() * 57344.0 - 1.0) < 1e-6 and abs(st[0].item() * st[1].item() - 1.0) < 1e-6
    # label entries of unconfident rows sit at (1 - p) ~ 1: exact in this scaling
    assert float(got.abs().max()) == 57344.0
    inplace = logits.clone()
    ops.ce_bwd(inplace, Vpad, V, Vpad, labels, mask, ls, lse, denom, rows)
    deq = q[:rows].float() * st[1]
    big = inplace.float().abs() > 1e-3 * float(inplace.float().abs().max())
    assert float(((deq - inplace.float()).abs() / inplace.float().abs().clamp_min(1e-30))[big].max()) < 0.14  # e5m2: 2 mantissa bits (+ bf16)
    wsum = gref.sum(0) / denom + 0.25
    assert float((flb[:V] - wsum).abs().max()) <= 1e-5 * float(wsum.abs().max()) + 1e-7 and float((flb[V:] - 0.25).abs().max()) == 0.0
    # label_coef: the label entries leave the byte matrix as fp32 coefficients; every other byte and the column sums stay
    q2 = torch.zeros((rows + 2, Vpad), dtype=E5, device=dev)
    coef = torch.full((rows,), 7.0, device=dev)
    flb2 = torch.full((Vpad,), 0.25, device=dev)
    ops.ce_bwd_q8(logits, Vpad, V, Vpad, labels, mask, ls, lse, denom, rows, ops.fp8_out(q2, st), colsum=flb2, label_coef=coef)
    torch.cuda.synchronize()
    ar = torch.arange(rows, device=dev)
    assert float(q2[ar, labels.long()].float().abs().max()) == 0.0
    q1 = q.clone()
    q1[ar, labels.long()] = 0
    assert torch.equal(q2.view(torch.uint8), q1.view(torch.uint8))
    assert float((flb2 - flb).abs().max()) <= 1e-5 * float(flb.abs().max())  # (fp32 atomics: the order of a column's 38 partial sums)
    cref = gref[ar, labels.long()] / denom
    assert float((coef - cref).abs().max()) <= 2e-6 * float(cref.abs().max())
    # a label that matches no column (an ignore index): the row's coefficient is cleared, not left as it was
    bad = labels.clone()
    bad[0], bad[rows - 1] = -100, V + 3
    coef2 = torch.full((rows,), 7.0, device=dev)
    ops.ce_bwd_q8(logits, Vpad, V, Vpad, bad, mask, ls, lse, denom, rows, ops.fp8_out(q2, st), label_coef=coef2)
    torch.cuda.synchronize()
    assert coef2[0].item() == 0.0 and coef2[rows - 1].item() == 0.0 and torch.equal(coef2[1:rows - 1], coef[1:rows - 1])


def test_head_label_terms(dev):
    from mic_amd import ops

    rows, width, V = 300, 1024, 5000
    g = torch.Generator().manual_seed(3)
    E = torch.randn(V, width, generator=g).to(torch.bfloat16).to(dev)
    h = torch.randn(rows + 4, width, generator=g).to(torch.bfloat16).to(dev)
    labels = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    labels[::5] = 2  # (eos: many rows meet in one gradient row)
    labels = labels.to(dev)
    coef = torch.randn(rows, generator=g).to(dev)
    coef[::7] = 0.0
    slab = torch.full((rows + 4, width), 3.0, device=dev)
    dE = torch.full((V, width), 0.5, device=dev)
    ops.head_label_terms(labels, coef, E, h, slab, dE, rows, width)
    torch.cuda.synchronize()
    want_slab = coef[:, None] * E[labels.long()].float()
    assert torch.equal(slab[:rows], want_slab) and float((slab[rows:] - 3.0).abs().max()) == 0.0
    want = torch.full((V, width), 0.5, device=dev, dtype=torch.float64)
    want.index_add_(0, labels.long(), (coef[:, None] * h[:rows].float()).double())
    assert float((dE.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("M", [2404, 300])
def test_fp8_head_projection_with_softmax_partials(dev, M):
    """MIC_FP8_HEAD=all: the LM-head forward on e4m3 operands carries the same by-product as the bf16 launch — (max, sum exp) of the
    stored logits per 64-column granule"""
    from mic_amd import ops

    N, K, nvalid = 4096, 1024, 4000
    g = torch.Generator().manual_seed(M)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(E4)
    b = (torch.randn(N, K, generator=g) * 0.5).to(dev).to(E4)
    sa, sb = torch.tensor([0.5], device=dev), torch.tensor([0.25], device=dev)
    bias = torch.randn(N, generator=g).to(dev)
    c = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
    stat = torch.zeros((M, 2 * (N // 64)), device=dev)
    ops.gemm(a, b, c, M, N, K, bias=bias, rowstat=stat, rowstat_nvalid=nvalid, a_scale_inv=sa, b_scale_inv=sb)
    want = (a.float() @ b.float().t()) * 0.125 + bias
    assert float((c.float() - want).abs().max()) <= 2e-2 * float(want.abs().max())
    x = c.float()
    x[:, nvalid:] = float("-inf")
    x = x.reshape(M, N // 64, 64)
    mx = x.max(dim=2).values
    sm = torch.exp(x - mx.clamp_min(-1e30)[..., None]).sum(dim=2)
    st = stat.reshape(M, N // 64, 2)
    live = mx > float("-inf")
    assert torch.equal(st[..., 0][live], mx[live])
    assert float(((st[..., 1] - sm)[live]).abs().max()) <= 1e-4 * float(sm[live].max())
