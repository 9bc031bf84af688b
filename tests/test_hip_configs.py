"""BASELINE.json's configurations at their REAL dimensions against the CPU oracle (oracle/reference_math.py, which
tests/test_oracle_golden.py pins to the reference's own outputs).

* C5 (Erdos-Renyi stress: D = d = 256, 4 GCN + 4 encoder layers, nhead 4 -> head_dim 64, ffn 1024, n = 512 + CLS,
  avg-deg 8): the whole model in the exact-fp32 mode at 1e-4, the encoder stack alone (fp32 at 1e-4, bf16 at 3e-2).
* C2 / C3 (Molpcba GIN-Virtual, Code2 GCN-Virtual: D = 300, d = 128, 128 / 5 x 5002 outputs): the FUSED path in the
  modes bench.py reports -- "mixed" (exact-fp32 MFMA for message passing, gnn2transformer and heads; bf16 token rows
  and bf16 MFMA inside the encoder layers) and "bf16" (bf16 MFMA everywhere, fp32 storage on the GNN side) -- against
  the fp32 oracle, with stated bounds on the loss and on every parameter gradient.
"""
from types import SimpleNamespace

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _args(**kw):
    a = dict(gnn_virtual_node=True, gnn_num_layer=5, gnn_emb_dim=300, gnn_JK="cat", gnn_dropout=0.0, gnn_residual=False,
             gnn_type="gcn", pretrained_gnn=None, freeze_gnn=None, d_model=128, nhead=4, dim_feedforward=512,
             transformer_dropout=0.0, transformer_activation="relu", num_encoder_layers=4, max_input_len=1000,
             transformer_norm_input=True, graph_pooling="cls", num_encoder_layers_masked=0, transformer_prenorm=False,
             pos_encoder=False, max_seq_len=5, compute_dtype=torch.float32, token_layout="auto")
    a.update(kw)
    return SimpleNamespace(**a)


ER_ARGS = dict(gnn_virtual_node=False, gnn_num_layer=4, gnn_emb_dim=256, gnn_JK="last", d_model=256, dim_feedforward=1024,
               max_seq_len=None)


def build(workload, args, graphs, seed):
    """model (CPU, fp32 parameters), batch (CPU), oracle loss function, HIP loss function"""
    from graphtrans_amd import losses, synth
    from graphtrans_amd.encoders import ASTNodeEncoder, AtomEncoder, BondEncoder
    from graphtrans_amd.models.gnn_transformer import GNNTransformer
    from oracle import reference_math as rm

    torch.manual_seed(seed)
    D = args.gnn_emb_dim
    if workload == "code2":
        model = GNNTransformer(5002, ASTNodeEncoder(D, 98, 10030, 20), lambda d: torch.nn.Linear(2, d), args)
        b = synth.code2_like(B=graphs, seed=seed)
        return model, b, (lambda out: rm.code2_loss(out, b.y_arr)), (lambda out, bd: losses.code2_loss(out, bd.y_arr))
    if workload == "molpcba":
        model = GNNTransformer(128, AtomEncoder(D), lambda d: BondEncoder(d), args)
        b = synth.molpcba_like(B=graphs, seed=seed)
        return model, b, (lambda out: rm.mol_loss(out, b.y)), (lambda out, bd: losses.mol_loss(out, bd.y))
    if workload == "er":
        model = GNNTransformer(2, torch.nn.Linear(256, D), lambda d: torch.nn.Linear(2, d), args)
        b = synth.er_stress(B=graphs, seed=seed)
        return model, b, (lambda out: rm.tud_loss(out, b.y)), (lambda out, bd: losses.tud_loss(out, bd.y))
    raise ValueError(workload)


def oracle_run(model, args, b, loss_of, dtype=torch.float32):
    """fp32 (the reference's arithmetic) or float64 (the conditioning-free answer of the same math) oracle run"""
    import copy

    from oracle import reference_math as rm

    sd = {k: (v.detach().to(dtype).requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    oargs = SimpleNamespace(**{k: v for k, v in vars(args).items() if k not in ("compute_dtype", "token_layout")})
    bb = copy.copy(b)
    for k in ("x", "edge_attr"):
        v = getattr(bb, k, None)
        if v is not None and v.is_floating_point():
            setattr(bb, k, v.to(dtype))
    torch.set_default_dtype(dtype)   # the oracle's helpers allocate zeros / ones in the default dtype
    try:
        out = rm.gnn_transformer(sd, oargs, bb, None, True)
        loss = loss_of(out)
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return out, loss.detach(), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


def fp32_noise(model, args, b, loss_of, ref64, eps=6e-8, seeds=(1, 2)):
    """Per gradient tensor: conftest.quantile_err of the fp32 ORACLE (torch CPU fp32 = the reference's arithmetic) against
    the float64 answer -- max over the unperturbed run and runs with every parameter moved by ~1 fp32 ulp, which sends
    the same math down different rounding paths.  At real sizes this noise is far above 1e-7 on some tensors: gradients
    of parameters in front of a train-mode BatchNorm are differences of large cancelling sums, and the gradient is a
    discontinuous function of the parameters (ReLU gates).  tests/test_hip_fp32_accuracy.py shows every HIP op on its
    own at fp32 roundoff; tools/flip_count.py shows the forward noise equal to torch's."""
    import copy

    from conftest import quantile_err

    noise = {k: 0.0 for k in ref64}
    for sd_ in (None,) + tuple(seeds):
        m2 = model
        if sd_ is not None:
            m2 = copy.deepcopy(model)
            g = torch.Generator().manual_seed(sd_)
            with torch.no_grad():
                for p in m2.parameters():
                    p.mul_(1 + eps * torch.randn(p.shape, generator=g))
        _, _, g32 = oracle_run(m2, args, b, loss_of, torch.float32)
        for k in noise:
            noise[k] = max(noise[k], quantile_err(g32[k], ref64[k]))
    return noise


TAIL_CAP = 5e-2


def check_grads(grads, ref64, noise, base=1e-3, factor=10.0, what=""):
    """per parameter-gradient tensor: 98 % of the elements within max(base, factor x the fp32 oracle's own noise) of the
    float64 oracle, relative to the tensor's largest entry (conftest.quantile_err: isolated gate flips are not counted),
    and the whole tensor within 5e-2 in relative L2 (gross errors).  base = 1e-3, not 1e-4: ONE flipped ReLU unit (a
    pre-activation within fp32 rounding of zero; a handful among the ~1e7 units of these models flip in any fp32
    evaluation, tools/flip_count.py) moves one row of an activation gradient by O(1e-2), and through the row sums every
    element of the LayerNorm / bias gradients above it by a few 1e-4 -- whether the fp32 oracle's three runs caught
    such a flip on the same tensor is chance.  The 1e-4 bar is enforced where it is meaningful at these sizes: on the
    loss and the logits (elementwise) here, and per op in tests/test_hip_fp32_accuracy.py (1e-6 in relative L2).  Tensors whose exact gradient is ~0 (a Linear bias
    in front of a train-mode BatchNorm: RMS below 1e-3 of the largest tensor RMS) are held to an absolute RMS bound of
    1e-5 of that scale instead."""
    from conftest import quantile_err, rel_l2

    rms = {k: float(r.norm()) / r.numel() ** 0.5 for k, r in ref64.items()}
    top = max(rms.values())
    rows = []
    for k, r in ref64.items():
        g = grads.get(k)
        g = torch.zeros_like(r) if g is None else g.double()
        if rms[k] < 1e-3 * top:
            e, tol = float((g - r).norm()) / r.numel() ** 0.5 / top, 1e-5
        else:
            e, tol = quantile_err(g, r), max(base, factor * noise[k])
            assert rel_l2(g, r) <= 5e-2, (k, rel_l2(g, r))
            # the 2 % tail the quantile leaves out is capped too: a gate flip moves an element by O(1e-2) of
            # the tensor's largest entry, a wrong row / tile by O(1)
            tail = float((g - r).abs().max()) / max(float(r.abs().max()), 1e-300)
            assert tail <= TAIL_CAP, (k, "excluded tail", tail)
        rows.append((e / tol, e, tol, k))
    rows.sort(reverse=True)
    msg = "; ".join(f"{k}: err {e:.1e} (tol {t:.1e})" for _, e, t, k in rows[:4])
    print(f"\n[{what}] gradients closest to their tolerance: {msg}")
    assert rows[0][0] <= 1.0, msg


def hip_run(model, b, hip_loss, matmul_dtype):
    from graphtrans_amd import ops

    ops.set_matmul_dtype(matmul_dtype)
    try:
        model = model.to(DEV).train()
        bd = b.to(DEV)
        out = model(bd)
        loss = hip_loss(out, bd)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
        outs = [o.detach().float().cpu() for o in (out if isinstance(out, (list, tuple)) else [out])]
        return outs, loss.detach().float().cpu(), grads
    finally:
        ops.set_matmul_dtype(torch.float32)


# --------------------------------------------------------------------------------------------------------------------
# C5: Erdos-Renyi stress configuration at its real dimensions
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False], ids=["engine", "modules"])
def test_c5_er_model_fp32_vs_oracle(fused):
    """GNNTransformer with the ER arguments (D = d = 256, 4 + 4 layers, nhead 4, ffn 1024) on 8 graphs of 512 nodes,
    avg-deg 8, exact-fp32 mode, against oracle.reference_math.gnn_transformer at 1e-4."""
    args = _args(**ER_ARGS)
    model, b, oloss, hloss = build("er", args, 8, 11)
    assert b.num_nodes == 8 * 512
    ref_out, ref_loss, ref_g = oracle_run(model, args, b, oloss)
    ref_out64, ref_loss64, ref_g64 = oracle_run(model, args, b, oloss, torch.float64)
    noise = fp32_noise(model, args, b, oloss, ref_g64)
    model.fused = fused
    outs, loss, grads = hip_run(model, b, hloss, torch.float32)
    assert abs(float(loss) - float(ref_loss64)) <= 1e-4 * max(1.0, abs(float(ref_loss64)))
    # logits are O(1): plain elementwise 1e-4 against both oracle precisions
    assert float((outs[0].double() - ref_out64.detach()).abs().max()) <= 1e-4
    assert float((outs[0] - ref_out.detach()).abs().max()) <= 1e-4
    from conftest import rel_l2
    print(f"\nlogits rel-L2 vs float64: HIP {rel_l2(outs[0], ref_out64.detach()):.1e}, fp32 oracle {rel_l2(ref_out.detach(), ref_out64.detach()):.1e}")
    check_grads(grads, ref_g64, noise, what="C5 fp32 " + ("engine" if fused else "modules"))


@pytest.mark.parametrize("mode", ["mixed", "bf16"])
def test_c5_er_model_reduced_precision_vs_oracle(mode):
    """The modes bench.py's ER line reports (only the fp32 mode was checked end to end): fused path, 8 graphs of
    512 nodes at the real dims, loss and logits at the mode's bound, every gradient tensor within LOWP_FACTOR x the
    oracle's own bf16-noise for that tensor (check_lowp_grads)."""
    from conftest import rel_l2

    matmul, tokens = MODES[mode]
    args = _args(**ER_ARGS, compute_dtype=tokens)
    model, b, oloss, hloss = build("er", args, 8, 11)
    ref_out64, ref_loss64, ref_g64 = oracle_run(model, args, b, oloss, torch.float64)
    outs, loss, grads = hip_run(model, b, hloss, matmul)
    rel = abs(float(loss) - float(ref_loss64)) / abs(float(ref_loss64))
    print(f"\n[er {mode}] loss rel err {rel:.2e}, logits rel-L2 {rel_l2(outs[0], ref_out64.detach()):.2e}")
    assert rel <= BOUNDS[mode]["loss"] * 4, rel
    assert rel_l2(outs[0], ref_out64.detach()) <= (2e-2 if mode == "mixed" else 5e-2)
    check_lowp_grads(model, args, b, oloss, grads, ref_g64, mode, what=f"er {mode}")


def _encoder_oracle(enc_state, args, x, mask, w, dtype, perturb_seed=None, eps=1e-7):
    from oracle import reference_math as rm

    sd = {k: v.detach().to(dtype).clone() for k, v in enc_state.items()}
    if perturb_seed is not None:
        g = torch.Generator().manual_seed(perturb_seed)
        sd = {k: (v.double() * (1 + eps * torch.randn(v.shape, generator=g, dtype=torch.float64))).to(dtype) for k, v in sd.items()}
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    xr = x.to(dtype).clone().requires_grad_(True)
    oargs = SimpleNamespace(**{k: v for k, v in vars(args).items() if k not in ("compute_dtype", "token_layout")})
    torch.set_default_dtype(dtype)
    try:
        ref, _ = rm.transformer_node_encoder(sd, "", oargs, xr, mask, True)
        (ref * w.to(dtype)).sum().backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return ref.detach(), xr.grad, {k: v.grad for k, v in sd.items() if v.grad is not None}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_c5_encoder_hd64_n513_vs_oracle(dtype):
    """TransformerNodeEncoder at d = 256, nhead 4 (head_dim 64), ffn 1024, 4 layers over sequences of 512 + CLS = 513
    positions (and ragged shorter ones), against oracle.transformer_node_encoder (= torch nn.TransformerEncoder math).
    fp32: outputs elementwise 1e-4, gradients: 98 % of each tensor's elements within max(1e-3, 10 x the fp32 oracle's own
    noise) (isolated ReLU gate flips excluded, conftest.quantile_err); bf16 token rows / bf16 MFMA: outputs
    3e-2 elementwise (scale-relative), gradients 8e-2 in relative L2."""
    from conftest import quantile_err, rel_l2
    from graphtrans_amd.modules.transformer_encoder import TransformerNodeEncoder

    torch.manual_seed(2)
    args = _args(**ER_ARGS, compute_dtype=dtype)
    enc = TransformerNodeEncoder(args)
    sizes = [512, 512, 300, 65, 512, 1]
    S, B, d = 512, len(sizes), 256
    x = torch.zeros(S, B, d)
    mask = torch.zeros(B, S, dtype=torch.bool)
    for i, n in enumerate(sizes):
        x[S - n:, i] = torch.randn(n, d)
        mask[i, :S - n] = True
    valid = torch.cat([~mask, torch.ones(B, 1, dtype=torch.bool)], 1).t().unsqueeze(-1)   # (S+1, B, 1): real positions
    w = torch.randn(S + 1, B, d) * valid     # the padded query rows carry no loss (their values are layout-dependent)
    state = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    ref, dx_ref, g_ref = _encoder_oracle(state, args, x, mask, w, torch.float64)
    enc = enc.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    out, _ = enc(xd, mask.to(DEV))
    (out.float() * w.to(DEV)).sum().backward()
    got = out.detach().float().cpu() * valid
    dx = xd.grad.cpu() * valid[:S]
    grads = {k: p.grad.float().cpu() for k, p in enc.named_parameters() if p.grad is not None}
    if dtype == torch.float32:
        assert_close(got, ref * valid, atol=1e-4, rtol=1e-4, what="encoder out")
        # fp32 noise of the reference arithmetic on the same tensors: fp32 oracle, unperturbed and moved by ~1 ulp
        noise = {}
        for seed in (None, 1, 2):
            _, dx32, g32 = _encoder_oracle(state, args, x, mask, w, torch.float32, perturb_seed=seed, eps=6e-8)
            noise["d x"] = max(noise.get("d x", 0.0), quantile_err(dx32 * valid[:S], dx_ref * valid[:S]))
            for k, r in g_ref.items():
                noise[k] = max(noise.get(k, 0.0), quantile_err(g32[k], r))
        errs = {"d x": quantile_err(dx, dx_ref * valid[:S])}
        errs.update({k: quantile_err(grads[k], r) for k, r in g_ref.items()})
        ratio = {k: errs[k] / max(1e-3, 10 * noise[k]) for k in errs}   # 1e-3: see check_grads
        worst = max(ratio, key=ratio.get)
        print(f"\n[C5 encoder fp32] worst gradient: {worst} 98%-quantile err {errs[worst]:.1e} (fp32 oracle noise {noise[worst]:.1e})")
        assert ratio[worst] <= 1.0, (worst, errs[worst], noise[worst])
        assert max(rel_l2(grads[k], r) for k, r in g_ref.items()) <= 5e-2
    else:
        assert_close(got, ref * valid, atol=3e-2, rtol=3e-2, what="encoder out")
        errs = {"d x": rel_l2(dx, dx_ref * valid[:S])}
        errs.update({k: rel_l2(grads[k], r) for k, r in g_ref.items()})
        worst = max(errs, key=errs.get)
        print(f"\n[C5 encoder bf16] worst relative L2 gradient error {errs[worst]:.2e} ({worst})")
        assert errs[worst] <= 8e-2, (worst, errs[worst])


# --------------------------------------------------------------------------------------------------------------------
# C2 / C3: the benchmarked precision modes of the fused path against the oracle, at the real dims
# --------------------------------------------------------------------------------------------------------------------
# Metrics (also carried in bench.py's JSON line as "precision_vs_oracle"): loss_rel_err = |loss - loss64| / |loss64|;
# per parameter-gradient tensor the relative L2 error ||g - g64|| / ||g64|| against the float64 oracle, over the tensors
# whose exact gradient is not ~0 (||g64|| >= 1e-3 of the largest tensor norm per element count; the Linear biases in
# front of a train-mode BatchNorm have an exactly-zero gradient).  "fp32" additionally goes through check_grads;
# the reduced-precision modes are held to the stated bounds on the worst and the median tensor.
LOWP_FACTOR, LOWP_FLOOR = 4.0, 2.5e-4   # floor: only a guard against a vanishing noise estimate (r4: the noise model rounds activations and their gradients too)
# elementwise logit error / largest logit: bf16 token rows round every stored activation to 2^-9 relative, four post-norm layers deep
LOGIT_MIXED, LOGIT_BF16 = 1.5e-2, 1e-1   # measured r5: mixed 5.7e-3 (Code2) / 4.6e-3 (Molpcba), bf16 4.2e-2 / 2.1e-2 -> ~2.5 x the measured value
# Since r5 the reduced-precision modes carry NO blanket bound on the worst / median gradient tensor (a blanket 0.2 / 0.8 can
# hide a wrong tensor): every tensor is held to its own oracle-noise bound (check_lowp_grads), the logits to an elementwise bound
# relative to the largest logit, and the same criterion runs on bench.py's own sample (test_bench_precision_sample_is_held_to_the_same_bound).
BOUNDS = {
    "fp32": dict(loss=1e-5, worst=2e-2, median=1e-3, logits=1e-4),
    "mixed": dict(loss=5e-4, logits=LOGIT_MIXED),
    "bf16": dict(loss=2e-3, logits=LOGIT_BF16),
}
MODES = {"fp32": (torch.float32, torch.float32), "mixed": (torch.float32, torch.bfloat16), "bf16": (torch.bfloat16, torch.bfloat16)}


@pytest.mark.parametrize("workload", ["code2", "molpcba"])
def test_bench_precision_sample_is_held_to_the_same_bound(workload):
    """bench.py's "precision_vs_oracle" sample (24 graphs, seed 7 / batch 5, dropout 0) through bench.py's OWN code, held to the bound
    of check_lowp_grads: every gradient tensor of the reduced-precision modes within LOWP_FACTOR x the oracle's bf16-noise of that
    tensor (the bench sample showed 7 x on GINConv.eps where this file's 64-graph sample passed at 4 x -- the noise
    estimate was a maximum over two perturbation seeds of a heavy-tailed response; it is taken over four seeds now, here and in
    bench.py, see oracle/noise.py)."""
    import bench

    rep = bench.precision_vs_oracle(workload, ["fp32", "mixed", "bf16"], DEV)
    for mode in ("mixed", "bf16"):
        r = rep[mode]
        print(f"\n[bench sample {workload} {mode}] loss rel err {r['loss_rel_err']:.2e}; worst tensor {r['grad_rel_l2_worst_param']} "
              f"{r['grad_rel_l2_worst']:.2e}; worst vs oracle noise {r['worst_vs_oracle_noise']} ({r['worst_vs_oracle_noise_param']})")
        assert r["loss_rel_err"] <= BOUNDS[mode]["loss"], r
        assert r["worst_vs_oracle_noise"] <= LOWP_FACTOR, r
    assert rep["fp32"]["loss_rel_err"] <= BOUNDS["fp32"]["loss"] and rep["fp32"]["grad_rel_l2_worst"] <= BOUNDS["fp32"]["worst"], rep["fp32"]


def precision_report(grads, loss, ref64, loss64):
    from conftest import rel_l2

    rms = {k: float(r.norm()) / r.numel() ** 0.5 for k, r in ref64.items()}
    top = max(rms.values())
    errs = {k: rel_l2(grads[k] if k in grads else torch.zeros_like(r), r) for k, r in ref64.items() if rms[k] >= 1e-3 * top}
    order = sorted(errs, key=errs.get, reverse=True)
    vals = sorted(errs.values())
    return dict(loss_rel_err=abs(float(loss) - float(loss64)) / abs(float(loss64)), grad_rel_l2_worst=errs[order[0]],
                grad_rel_l2_worst_param=order[0], grad_rel_l2_median=vals[len(vals) // 2], tensors=len(errs),
                tensors_with_zero_gradient=len(ref64) - len(errs), worst4=[(k, errs[k]) for k in order[:4]])


@pytest.mark.parametrize("mode", ["fp32", "mixed", "bf16"])
@pytest.mark.parametrize("workload,graphs", [("code2", 24), ("molpcba", 64)])
def test_fused_precision_modes_vs_oracle(workload, graphs, mode):
    from graphtrans_amd import engine, ops

    matmul, tokens = MODES[mode]
    kw = dict(compute_dtype=tokens)
    if workload == "molpcba":
        kw.update(gnn_type="gin", max_seq_len=None)
    args = _args(**kw)
    model, b, oloss, hloss = build(workload, args, graphs, 5)
    ref_out64, ref_loss64, ref_g64 = oracle_run(model, args, b, oloss, torch.float64)
    noise = fp32_noise(model, args, b, oloss, ref_g64) if mode == "fp32" else None
    o32 = None
    if mode == "fp32":   # (before the model moves to the GPU: the oracle runs on CPU tensors)
        _, o32_loss, o32_g = oracle_run(model, args, b, oloss, torch.float32)
        o32 = precision_report(o32_g, o32_loss, ref_g64, ref_loss64)
    ops.set_matmul_dtype(matmul)
    try:
        assert engine.eligible(model.to(DEV).train(), b.to(DEV), None), "the benchmarked configuration must run on the fused path"
    finally:
        ops.set_matmul_dtype(torch.float32)
    outs, loss, grads = hip_run(model, b, hloss, matmul)
    rep = precision_report(grads, loss, ref_g64, ref_loss64)
    print(f"\n[{workload} {mode}] loss {float(loss):.6f} vs fp64 oracle {float(ref_loss64):.6f} (rel {rep['loss_rel_err']:.2e}); "
          f"grad rel-L2 err worst {rep['grad_rel_l2_worst']:.2e} median {rep['grad_rel_l2_median']:.2e} over {rep['tensors']} tensors; "
          "worst: " + ", ".join(f"{k} {e:.1e}" for k, e in rep["worst4"]))
    bound = BOUNDS[mode]
    assert rep["loss_rel_err"] <= bound["loss"], rep
    # the logits, element by element, relative to the largest logit (they are O(1..10): the fp32 mode is the plain 1e-4)
    refs = [o.detach() for o in (ref_out64 if isinstance(ref_out64, (list, tuple)) else [ref_out64])]
    top = max(float(r.abs().max()) for r in refs)
    lerr = max(float((o.double() - r).abs().max()) for o, r in zip(outs, refs)) / max(top, 1.0)
    print(f"[{workload} {mode}] logits: max elementwise error {lerr:.2e} of the largest logit ({top:.2f}); bound {bound['logits']:g}")
    assert lerr <= bound["logits"], (lerr, bound["logits"])
    if mode == "fp32":
        assert rep["grad_rel_l2_worst"] <= bound["worst"], rep
        # the median tensor: 1e-3, or 3 x the median of the fp32 ORACLE itself against float64 on this batch where that is larger (r6: Molpcba's
        # 64-graph sample sits AT 1e-3 for any fp32 evaluation -- GINConv.eps and the BatchNorm-fronted weights are differences of large
        # cancelling sums; another summation order in the virtual-node MLP moved the HIP path from just below to 1.05e-3)
        print(f"[{workload} fp32] the fp32 oracle itself: grad rel-L2 err worst {o32['grad_rel_l2_worst']:.2e} median {o32['grad_rel_l2_median']:.2e}")
        assert rep["grad_rel_l2_median"] <= max(bound["median"], 3.0 * o32["grad_rel_l2_median"]), (rep, o32)
        check_grads(grads, ref_g64, noise, what=f"{workload} fp32")
    else:
        check_lowp_grads(model, args, b, oloss, grads, ref_g64, mode, what=f"{workload} {mode}")




def check_lowp_grads(model, args, b, oloss, grads, ref_g64, mode, what=""):
    """Reduced-precision modes, tensor by tensor: the relative L2 error of every parameter gradient against the float64
    oracle is bounded by LOWP_FACTOR x the ORACLE's own response of that tensor to bf16-sized perturbations of the GEMM
    weights AND of the activations / activation gradients the mode rounds (oracle/noise.py: 2^-9 relative, through reference_math's
    storage taps), never less than LOWP_FLOOR (a guard against a vanishing estimate).  An ill-conditioned gradient (GINConv.eps: one scalar summed
    from N x D products of both signs, modules/conv.py:21,28) gets a wide bound because the oracle itself moves that
    much, every well-conditioned tensor a tight one -- instead of one blanket bound per mode."""
    from conftest import rel_l2
    from oracle import noise as on
    from oracle import reference_math as rm

    sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}   # (hip_run moved the model)
    floor = on.lowp_noise(sd64, on.oracle_args(args), b, rm.gnn_transformer, oloss, ref_g64, mode)
    rms = {k: float(r.norm()) / r.numel() ** 0.5 for k, r in ref_g64.items()}
    top = max(rms.values())
    rows = []
    for k, r in ref_g64.items():
        if rms[k] < 1e-3 * top:
            continue   # exact gradient ~ 0 (a bias in front of a train-mode BatchNorm): covered by the fp32 mode's absolute bound
        e = rel_l2(grads[k] if k in grads else torch.zeros_like(r), r)
        tol = max(LOWP_FACTOR * floor[k], LOWP_FLOOR)
        rows.append((e / tol, e, floor[k], k))
    rows.sort(reverse=True)
    msg = "; ".join(f"{k}: err {e:.1e} = {e / max(f, 1e-30):.1f} x oracle noise {f:.1e}" for _, e, f, k in rows[:5])
    print(f"\n[{what}] gradients closest to their bound ({LOWP_FACTOR:g} x oracle bf16-noise, floor {LOWP_FLOOR:g}): {msg}")
    assert rows[0][0] <= 1.0, msg
