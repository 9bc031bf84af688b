#!/usr/bin/env python
"""bench.py — GraphTrans training-step throughput on MI355X (driver contract).

    python bench.py [--gpus N --steps K --warmup W] [--workload code2|molpcba|nci1|er|code2-pna] [--mode mixed|bf16|fp32]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under `torch.distributed.run` with N ranks
(one per GPU, RCCL); the driver's own `python -m torch.distributed.run ... bench.py --gpus N` works as before.

Metric (BASELINE.json): graphs/sec (fwd+bwd) OGBG-Code2 GCN-Virtual b256.  One "step" = the whole hot path over one
synthetic, HBM-resident, pre-collated batch of 256 graphs PER GPU (weak scaling: the graphs are independent units, no
data-path collective):
    zero grads -> graph_prep -> GNNTransformer forward -> loss (dataset/code.py:39-45) -> backward
    -> [RCCL gradient all-reduce, overlapped] -> fused AdamW step
Nothing is skipped or cached across steps (the per-batch graph structure is rebuilt every step; dropout runs at the
reference's configured rates).  Inputs rotate over 4 seeded batches.

Precision modes (`config.mode`):
  mixed (default, the headline `value`): fp32-accurate arithmetic for message passing, the virtual-node MLPs, gnn2transformer
         and the prediction heads -- the big-M linears as six bf16 products per fp32 product on the bf16 matrix pipe (csrc/linear3x.h:
         relative error to float64 like torch's fp32 GEMM; GT_F32_GEMM=exact selects v_mfma_f32_16x16x4_f32 everywhere), the short-M
         ones on exact-fp32 MFMA; bf16 token rows and bf16 MFMA inside the encoder layers (attention / in- and out-projection / FFN)
         -- what BASELINE.json's north_star sanctions;
  bf16:  bf16 MFMA for every GEMM (fp32 storage and master weights on the GNN side);
  fp32:  fp32-accurate GEMMs everywhere (fp32 token rows).
Rank 0 prints ONE JSON line of at most 4 KB (compact_line) with the contract keys plus
  "roofline":     the dominant hand-written kernel, named as rocprofv3 names it (the kernel family with the largest total
                  HIP-event time inside the timed region; forward and dX launches of one kernel are one family): algorithmic
                  bytes|flops per launch (SURVEY.md 8d formulas) / average launch time; "traffic" = PMC bytes of the same
                  build (profiles/*_pmc_traffic.json, attached only when its build id matches the sources that run),
  "cpu_baseline": the CPU oracle (oracle/reference_math.py, kind "port") timed on this box's host cores on a bounded
                  sample of the same workload (rank 0, N = 1 only),
  "value_fp32_contract" / "ms_per_step_fp32_contract": the same step in the fp32 mode (the mode that meets north_star's
                  1e-4), "value_bf16": the all-bf16 mode, "precision_vs_oracle": three numbers per mode,
  "strong_scaling" / "weak_scaling": (N > 1) value + ms_per_step of the other scaling of the same step,
  "report":       path of bench_report.json -- the FULL record (every timed kernel's roofline, the other modes' kernels,
                  aggregate_stress, collate, the precision report's details, notes), written next to this script.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PROF_MASK = 1 | 2 | 32       # csrc/gt_common.h: aggregate + attention entry points, the GEMM KERNELS one by one on their launch streams
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
MFMA_F32_PEAK_TF = 157.3     # fp32-input MFMA
MFMA_BF16X6_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0   # fp32-accurate products on the bf16 pipe: six bf16 MFMAs each (csrc/linear3x.h)


def model_args(workload, dtype):
    """Reference hyper-parameters (SURVEY.md §8a sizes; main.py:53-58, transformer_encoder.py:13-20,
    configs/code2/gnn-transformer/JK=cat/pooling=cls+norm_input.yml, configs/molpcba/...+gin+norm_input.yml)."""
    from types import SimpleNamespace

    a = dict(gnn_virtual_node=True, gnn_num_layer=5, gnn_emb_dim=300, gnn_JK="cat", gnn_dropout=0.0,
             gnn_residual=False, gnn_type="gcn", pretrained_gnn=None, freeze_gnn=None, d_model=128, nhead=4,
             dim_feedforward=512, transformer_dropout=0.3, transformer_activation="relu", num_encoder_layers=4,
             max_input_len=1000, transformer_norm_input=True, graph_pooling="cls", num_encoder_layers_masked=0,
             transformer_prenorm=False, pos_encoder=False, max_seq_len=5, compute_dtype=dtype, token_layout="auto")
    if workload == "molpcba":
        a.update(gnn_type="gin", gnn_dropout=0.3, max_seq_len=None)
    elif workload == "nci1":
        a.update(gnn_virtual_node=False, gnn_num_layer=3, gnn_emb_dim=128, gnn_JK="last", gnn_dropout=0.5,
                 dim_feedforward=256, transformer_dropout=0.1, num_encoder_layers=3, transformer_norm_input=False,
                 max_seq_len=None)
    elif workload == "er":
        a.update(gnn_virtual_node=False, gnn_num_layer=4, gnn_emb_dim=256, gnn_JK="last", d_model=256,
                 dim_feedforward=1024, transformer_dropout=0.0, max_seq_len=None)
    elif workload == "code2-pna":  # configs/code2/pna-transformer/pooling=cls+norm_input.yml:16-23
        a.update(gnn_virtual_node=False, gnn_num_layer=4, gnn_emb_dim=272, gnn_JK="last", gnn_residual=True,
                 gnn_dropout=0.0, aggregators=["mean", "max", "min", "std"],
                 scalers=["identity", "amplification", "attenuation"], deg=torch.tensor([0, 4000, 2500, 900, 300, 90, 30, 9]))
    if os.environ.get("GT_BENCH_TDROP") is not None:  # ablation only: transformer dropout rate
        a["transformer_dropout"] = float(os.environ["GT_BENCH_TDROP"])
    return SimpleNamespace(**a)


def build(workload, dtype, device, batch_graphs):
    from graphtrans_amd import losses, synth
    from graphtrans_amd.encoders import ASTNodeEncoder, AtomEncoder, BondEncoder
    from graphtrans_amd.models.gnn_transformer import GNNTransformer

    args = model_args(workload, dtype)
    D = args.gnn_emb_dim
    if workload == "code2":
        model = GNNTransformer(5002, ASTNodeEncoder(D, 98, 10030, 20), lambda d: torch.nn.Linear(2, d), args)
        gen = lambda seed: synth.code2_like(B=batch_graphs, seed=seed)
        loss = lambda out, b: losses.code2_loss(out, b.y_arr)
        name = "OGBG-Code2-like synthetic, GraphTrans GCN-Virtual L5 D300 JK=cat cls norm_input, 4 enc layers d128"
    elif workload == "molpcba":
        model = GNNTransformer(128, AtomEncoder(D), lambda d: BondEncoder(d), args)
        gen = lambda seed: synth.molpcba_like(B=batch_graphs, seed=seed)
        loss = lambda out, b: losses.mol_loss(out, b.y)
        name = "OGBG-Molpcba-like synthetic, GraphTrans GIN-Virtual L5 D300 JK=cat cls norm_input"
    elif workload == "code2-pna":
        from graphtrans_amd.models.pna_transformer import PNATransformer
        model = PNATransformer(5002, ASTNodeEncoder(D, 98, 10030, 20), None, args)
        gen = lambda seed: synth.code2_like(B=batch_graphs, seed=seed)
        loss = lambda out, b: losses.code2_loss(out, b.y_arr)
        name = "OGBG-Code2-like synthetic, GraphTrans (PNA) L4 D272 towers 4, cls norm_input, 4 enc layers d128"
    elif workload == "nci1":
        def zero_cls(_):
            return lambda _x: 0
        model = GNNTransformer(2, torch.nn.Linear(37, D), zero_cls, args)
        gen = lambda seed: synth.nci1_like(B=batch_graphs, seed=seed)
        loss = lambda out, b: losses.tud_loss(out, b.y)
        name = "NCI1-like synthetic, GraphTrans(small, GCN) d128, 3 GNN layers"
    elif workload == "er":
        model = GNNTransformer(2, torch.nn.Linear(256, D), lambda d: torch.nn.Linear(2, d), args)
        gen = lambda seed: synth.er_stress(B=batch_graphs, seed=seed)
        loss = lambda out, b: losses.tud_loss(out, b.y)
        name = "Erdos-Renyi G(512, 8/511) stress, 4 GCN + 4 encoder layers d256"
    else:
        raise ValueError(workload)
    return args, model.to(device), gen, loss, name


RAW_GEN = {"code2": "code2_raw", "code2-pna": "code2_raw", "molpcba": "molpcba_raw"}   # synth generators


def make_store(workload, num_graphs, seed):
    from graphtrans_amd.data import GraphStore
    if workload not in RAW_GEN:
        raise SystemExit("--from-store: no raw graph store for workload %r" % workload)
    from graphtrans_amd import synth
    return GraphStore(getattr(synth, RAW_GEN[workload])(B=num_graphs, seed=seed))


def collate_report(workload, per_gpu, with_cpu):
    """Batch assembly beside the step (collation is reported separately): device = gt_collate from the
    HBM store (sample -> augment_edge -> concatenate), cpu = the numpy restatement of the reference's per-sample
    transform + PyG collation (oracle/collate.py) on the same graphs, one thread."""
    from graphtrans_amd import synth
    from graphtrans_amd.data import GraphStore
    raw = getattr(synth, RAW_GEN[workload])(B=2 * per_gpu, seed=77)
    store = GraphStore(raw)
    rng = np.random.default_rng(0)
    ids = [rng.permutation(len(raw))[:per_gpu] for _ in range(20)]
    for i in ids[:3]:
        store.collate(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in ids:
        store.collate(i)
    torch.cuda.synchronize()
    rep = {"device_us_per_batch": round((time.perf_counter() - t0) / len(ids) * 1e6, 1), "graphs": per_gpu,
           "store_mb": round(store.nbytes() / 1e6, 1)}
    if with_cpu:
        from oracle import collate as oc
        t0 = time.perf_counter()
        for i in ids[:3]:
            oc.collate([raw[j] for j in i], augment=store.augment)
        rep["cpu_us_per_batch"] = round((time.perf_counter() - t0) / 3 * 1e6, 1)
    return rep


def aggregate_stress_report(device):
    """The aggregate kernels' bandwidth proof (on Code2 the re-gathers hit L2, so its fractions are cache numbers):
    gt_aggregate_fwd / _bwd alone on one batch of BASELINE configs[4] (256 x G(512, 8/511): 131 k nodes, ~1.05 M edges, D = 256,
    GCN with the Linear(2) edge encoder), 20 launches each timed by the C-side launch profiler (HIP events on the launch stream)."""
    from graphtrans_amd import _lib, ops, synth
    from graphtrans_amd.graph import GraphStructure
    from graphtrans_amd.modules.conv import edge_spec

    D = 256
    b = synth.er_stress(B=256, seed=0).to(device)
    gs = GraphStructure.build(b.edge_index, b.batch, num_graphs=b.num_graphs)
    torch.manual_seed(0)
    lin = torch.nn.Linear(2, D).to(device)
    h = torch.randn(gs.N, D, device=device, requires_grad=True)
    root = torch.randn(1, D, device=device, requires_grad=True)
    g = torch.randn(gs.N, D, device=device)
    spec = edge_spec(lin, b.edge_attr, D)
    for it in range(25):
        if it == 5:
            _lib.profile_enable(1)
        ops.aggregate(h, gs, "gcn", root, spec).backward(g)
    rec = _lib.profile_records()
    _lib.profile_enable(0)
    meta = dict(N=gs.N, E=gs.E, D=D, elt=4, attr_bytes=8)
    out = {"workload": "BASELINE configs[4] batch: 256 x G(512, 8/511)", "N": int(gs.N), "E": int(gs.E), "D": D, "dtype": "f32"}
    for name, bwd in (("gt_aggregate_fwd", False), ("gt_aggregate_bwd", True)):
        ms = [m for n, m, _ in rec if n == name]
        if not ms:
            continue
        us = 1e3 * sum(ms) / len(ms)
        by = agg_bytes(meta, bwd)
        out[name] = {"bound": "hbm", "avg_us": round(us, 1), "calls": len(ms), "algorithmic_bytes": int(by),
                     "achieved": round(by / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / us / 1e3 / HBM_PEAK_GBS, 4),
                     "traffic": None}
    # HBM-side bytes per launch from the PMC passes over tools/agg_stress.py (tools/pmc_round.sh), same-build files only
    try:
        bid = build_id()
        pdir = os.path.join(REPO, "profiles")
        for fn in sorted(os.listdir(pdir), reverse=True):
            if fn.endswith("_aggregate_stress_pmc_traffic.json"):
                d = json.load(open(os.path.join(pdir, fn)))
                if d.get("build_id") == bid:
                    for k, v in d.get("traffic", {}).items():
                        if k in out:
                            out[k]["traffic"] = v
                            out[k]["traffic_frac_of_peak"] = round(v / (out[k]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    out["traffic_source"] = "profiles/%s (build %s)" % (fn, bid)
                    break
    except OSError:
        pass
    return out


def attach_sizes(b):
    b._sizes = torch.bincount(b.batch, minlength=b.num_graphs).numpy()
    return b


# ---- algorithmic work per launch ---------------------------------------------------
def agg_bytes(meta, bwd):
    N, E, D, s, a = meta["N"], meta["E"], meta["D"], meta["elt"], meta["attr_bytes"]
    if not bwd:   # E*8 + (N+1)*8 + E*a + E*D*s + N*D*s + N*4 + N*D*s
        return E * 8 + (N + 1) * 8 + E * a + E * D * s + N * D * s + N * 4 + N * D * s
    return E * 8 + (N + 1) * 8 + E * a + E * D * s + 2 * N * D * s + N * D * s


def build_id():
    """sha256 over the sources that decide what runs on the GPU (kernels, C-ABI, the fused path): PMC traffic files carry
    the id of the build they were measured on, and are attached only to runs of the same build (no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    roots = [os.path.join(REPO, "graphtrans_amd", "csrc"), os.path.join(REPO, "include")]
    files = sorted(os.path.join(r, f) for r in roots for f in os.listdir(r) if f.endswith((".hip", ".h")))
    files += [os.path.join(REPO, "graphtrans_amd", f) for f in ("engine.py", "layers.py", "ops.py", "graph.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, mode, per_gpu):
    """-> (traffic dict, note).  The newest profiles/*_pmc_traffic.json of this workload / mode / batch whose build id is
    the build that is running; a file from another build is NOT attached (note says so)."""
    pdir = os.path.join(REPO, "profiles")
    bid, stale = build_id(), None
    try:
        for fn in sorted(os.listdir(pdir), reverse=True):
            if not fn.endswith("_pmc_traffic.json"):
                continue
            d = json.load(open(os.path.join(pdir, fn)))
            if d.get("workload") != workload or d.get("mode", d.get("dtype")) != mode or d.get("graphs_per_gpu") != per_gpu:
                continue
            if d.get("build_id") == bid:
                return d.get("traffic", {}), "profiles/%s (build %s)" % (fn, bid)
            stale = stale or fn
    except OSError:
        pass
    return {}, ("not attached: profiles/%s was measured on another build (running build %s)" % (stale, bid)) if stale \
        else "no PMC profile for this workload / mode (running build %s)" % bid


def _w3_enabled():
    from graphtrans_amd import w3
    return bool(w3.ENABLED)


POOLED = ("k_lin3r", "k_lin1")   # one template instantiation serves forward and dX: ONE kernel name in rocprofv3's table


def pooled_name(name):
    """`k_lin3r[fwd]` / `k_lin3r[dx]` -> `k_lin3r` (rocprofv3: k_lin3r<10,2> for both); `k_lin1[fwd]` / `k_lin1[dx]` -> `k_lin1`;
    the epilogue variants (`[fwd+ln]`, `[dx+lnb]`: other template modes) and every other kernel keep their names"""
    for k in POOLED:
        for d in ("[fwd]", "[dx]"):
            if name == k + d:
                return k
    return name


def kernel_report(records, attn_flops_fwd, dtype, matmul_dtype=None, pool=False):
    """pool=True: forward and dX launches of one kernel instantiation form one entry (pooled_name) -- the table the
    dominant kernel of the line's "roofline" is picked from, comparable with rocprofv3's per-kernel rows.
    records: [(name, ms, dims6)] from the C-side launch profiler (HIP events on the stream each launch goes to, recorded
    inside the timed region, under the schedule the un-profiled steps run: the weight-gradient GEMMs are bracketed ON the
    overlap stream).  -> per kernel roofline dicts, keyed like rocprofv3's kernel names so that every `frac` can be
    recomputed from profiles/*_kernel_stats.csv: achieved = algorithmic flops | bytes / average duration.
      k_lin3[fwd|dx], k_lin3r[fwd|dx] (rows straight into fragments, r5), k_lin3_dw, k_lin3r_dw (the GEMM kernel alone since r6: its reduce is k_split_reduce's own rocprofv3 row)   fp32-accurate GEMM on the bf16 pipe, six bf16 MFMAs per product: peak = 2500 / 6 = 416.7 TFLOP/s
      k_lin32[..], k_lin32_dw   exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): peak 157.3 TFLOP/s
      k_linear_*          bf16 MFMA on skinny shapes (M ~ 3e4, K, N <= 600): HBM-bound, priced on algorithmic bytes
      k_small_*           short-M GEMMs (one row per graph): latency-bound, reported against the fp32 MFMA peak for scale"""
    groups = {}
    for name, ms, dims in records:
        if pool:
            name = pooled_name(name)
        if name.startswith("k_"):
            tag = "" if name.startswith(("k_lin3", "k_lin32")) else ("[fp32" if dims[5] == 0 else "[bf16") + (",rows=graphs]" if dims[0] <= 1024 else "]")
            name = name + tag
        groups.setdefault(name, []).append((ms, dims))
    rep = {}
    for name, items in groups.items():
        calls = len(items)
        total_ms = sum(ms for ms, _ in items)
        avg_us = 1e3 * total_ms / calls
        base = dict(avg_us=round(avg_us, 2), calls=calls, total_ms=round(total_ms, 3), traffic=None)
        if name.startswith("gt_aggregate"):
            bwd = name.endswith("bwd")
            per = float(np.mean([agg_bytes(dict(N=d[0], E=d[1], D=d[2], elt=d[3], attr_bytes=d[4]), bwd) for _, d in items]))
            gbs = per / (avg_us * 1e-6) / 1e9
            rep[name] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(gbs / HBM_PEAK_GBS, 4), algorithmic_bytes=int(per), **base)
        elif name.startswith("gt_attn"):
            per = attn_flops_fwd * (2.5 if name.endswith("bwd") else 1.0)
            tf = per / (avg_us * 1e-6) / 1e12
            peak = MFMA_BF16_PEAK_TF if dtype == torch.bfloat16 else MFMA_F32_PEAK_TF
            rep[name] = dict(bound="mfma", achieved=round(tf, 2), peak=peak, unit="TFLOP/s", frac=round(tf / peak, 5),
                             algorithmic_flops=int(per), **base)
        elif name.startswith("k_"):
            kind = name.split("[")[0]

            def lin_bytes(d):
                M, N, K, xd, yd, _ = d
                ex, ey = (2 if xd == 1 else 4), (2 if yd == 1 else 4)
                return M * K * ex + M * N * ey + N * K * 4
            per = float(np.mean([lin_bytes(d) for _, d in items]))
            gbs = per / (avg_us * 1e-6) / 1e9
            fl = float(np.mean([2.0 * d[0] * d[1] * d[2] for _, d in items]))
            tf = fl / (avg_us * 1e-6) / 1e12
            if kind.startswith("k_lin3") and not kind.startswith("k_lin32"):
                rep[name] = dict(bound="mfma", achieved=round(tf, 2), peak=round(MFMA_BF16X6_PEAK_TF, 1), unit="TFLOP/s",
                                 frac=round(tf / MFMA_BF16X6_PEAK_TF, 4), algorithmic_flops=int(fl), algorithmic_bytes=int(per), gbs=round(gbs, 1),
                                 peak_note="dense bf16 MFMA 2500 TFLOP/s / 6 bf16 products per fp32-accurate product", **base)
            elif kind.startswith("k_lin32") or "[fp32" in name:
                rep[name] = dict(bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                                 frac=round(tf / MFMA_F32_PEAK_TF, 4), algorithmic_flops=int(fl), algorithmic_bytes=int(per),
                                 gbs=round(gbs, 1), **base)
            else:
                rep[name] = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=round(gbs / HBM_PEAK_GBS, 4), algorithmic_bytes=int(per), tflops=round(tf, 1), **base)
    return rep


# ---- CPU baseline (the oracle as a timed port) --------------------------------------------------
def _oracle_case(workload, graphs):
    from graphtrans_amd import synth
    from oracle import reference_math as rm
    if workload == "code2":
        b = synth.code2_like(B=graphs, seed=0)
        return b, rm.gnn_transformer, (lambda out: rm.code2_loss(out, b.y_arr)), "Code2-like"
    if workload == "molpcba":
        b = synth.molpcba_like(B=graphs, seed=0)
        return b, rm.gnn_transformer, (lambda out: rm.mol_loss(out, b.y)), "Molpcba-like"
    if workload == "code2-pna":
        b = synth.code2_like(B=graphs, seed=0)
        return b, rm.pna_transformer, (lambda out: rm.code2_loss(out, b.y_arr)), "Code2-like (PNA)"
    if workload == "nci1":   # BASELINE configs[0]: the reference's own CPU-runnable case
        b = synth.nci1_like(B=graphs, seed=0)
        return b, rm.gnn_transformer, (lambda out: rm.tud_loss(out, b.y)), "NCI1-like"
    if workload == "er":
        b = synth.er_stress(B=graphs, seed=0)
        return b, rm.gnn_transformer, (lambda out: rm.tud_loss(out, b.y)), "Erdos-Renyi stress"
    raise ValueError(workload)


def _time_oracle(sd, args, b, fwd, loss_of, threads, budget_s, min_iters=1):
    """median seconds of forward + loss + backward; one untimed warm-up when the budget allows it"""
    torch.set_num_threads(threads)
    times, t_start = [], time.perf_counter()
    it = 0
    while True:
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss_of(fwd(sd, args, b, None, True)).backward()
        dt = time.perf_counter() - t0
        if it > 0 or dt > budget_s / 2:   # a slow point: the first iteration is the sample
            times.append(dt)
        it += 1
        spent = time.perf_counter() - t_start
        if (len(times) >= min_iters and spent + dt > budget_s) or len(times) >= 7:
            break
    return float(np.median(times)), len(times)


CPU_SAMPLE = {"code2": 64, "code2-pna": 64, "molpcba": 256, "nci1": 256, "er": 16}


def cpu_baseline(workload, model, args, budget_s=24.0, full_graphs=256):
    """The oracle timed as a CPU port, bounded to ~budget_s of CPU work so that the default run
    stays within minutes: the best multi-thread point (32 threads: torch's CPU kernels get SLOWER with more on these
    many small ops -- 256 threads: 154 s per 64-graph step; tools/cpu_sweep.py, profiles/r02_cpu_sweep.json) and one
    thread (the per-core figure), both on a 64-graph sample.  The full protocol (256 graphs, every thread count up to os.cpu_count(), median of 5)
    is `python bench.py --cpu-baseline-full` (minutes of CPU time; its output is committed under profiles/)."""
    from types import SimpleNamespace
    graphs = CPU_SAMPLE[workload]
    b, fwd, loss_of, what = _oracle_case(workload, graphs)
    oargs = SimpleNamespace(**{k: v for k, v in vars(args).items() if k not in ("compute_dtype", "token_layout")})
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cores = os.cpu_count() or 1
    threads = min(32, cores)
    t, n = _time_oracle(sd, oargs, b, fwd, loss_of, threads, 0.55 * budget_s)
    g1 = graphs
    t1, n1 = _time_oracle(sd, oargs, b, fwd, loss_of, 1, 0.45 * budget_s)
    full = None
    if graphs < full_graphs and not os.environ.get("GT_BENCH_NO_CPU_FULL"):
        # the batch the GPU line is quoted on (the sample flatters the CPU -- the padded attention of the
        # reference layout grows with the longest graph of the batch): ONE timed pass, no warm-up (34 s at 32 threads)
        bf, fwd_f, loss_f, _ = _oracle_case(workload, full_graphs)
        tf, nf = _time_oracle(sd, oargs, bf, fwd_f, loss_f, threads, 1.0)
        full = dict(value=round(full_graphs / tf, 2), unit="graphs/s", cores=threads, graphs=full_graphs, iters=nf, s_per_step=round(tf, 3),
                    note="the full batch of the GPU line, one timed pass without warm-up")
    torch.set_num_threads(threads)
    return dict(value=round(graphs / t, 2), unit="graphs/s", cores=threads, kind="port", full_batch=full,
                sample_short=f"oracle/reference_math.py fwd+loss+bwd fp32, {graphs}-graph seed-0 {what} batch, median of {n} iter(s), {threads} of {cores} threads",
                sample=f"oracle/reference_math.py fwd+loss+bwd fp32 (dropout at config rates) on a {graphs}-graph seed-0 {what} "
                       f"batch, median of {n} timed iteration(s) on {threads} threads; padded layout like the reference (S = max "
                       f"nodes of the sample); host has {cores} logical cores (more threads are slower: profiles/r02_cpu_sweep.json)",
                s_per_step=round(t, 3),
                one_thread=dict(value=round(g1 / t1, 2), unit="graphs/s", cores=1, graphs=g1, iters=n1, s_per_step=round(t1, 3)))


def cpu_baseline_full(workload, model, args, graphs=256, iters=5, limit_s=900.0):
    """BASELINE.md section 3 in full: `graphs` graphs, 2 warm-ups + `iters` timed iterations (median) per thread count,
    thread counts 1 .. os.cpu_count() in powers of two; a point that would exceed limit_s is cut short.  The 256-graph
    batch costs 34-60 s per iteration (229 s on all 256 threads): ~1 h in full, which is why the default run is a sample;
    profiles/r02_cpu_sweep.json holds one full sweep of this box (7.5 graphs/s at 32 threads, 4.3 at one)."""
    from types import SimpleNamespace
    b, fwd, loss_of, what = _oracle_case(workload, graphs)
    oargs = SimpleNamespace(**{k: v for k, v in vars(args).items() if k not in ("compute_dtype", "token_layout")})
    sd = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cores = os.cpu_count() or 1
    counts = sorted({1, cores} | {2 ** i for i in range(1, 12) if 2 ** i <= cores})
    points = []
    for th in counts:
        torch.set_num_threads(th)
        ts, t_start = [], time.perf_counter()
        for it in range(2 + iters):
            for v in sd.values():
                v.grad = None
            t0 = time.perf_counter()
            loss_of(fwd(sd, oargs, b, None, True)).backward()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > limit_s:
                break
        timed = ts[2:] if len(ts) > 2 else ts[-1:]
        points.append(dict(threads=th, s_per_step=round(float(np.median(timed)), 3), graphs_per_s=round(graphs / float(np.median(timed)), 2),
                           timed_iterations=len(timed)))
        print("cpu_baseline_full:", points[-1], file=sys.stderr, flush=True)
    best = max(points, key=lambda p: p["graphs_per_s"])
    return dict(workload=what, graphs=graphs, nodes=int(b.num_nodes), cpu_count=cores, points=points, best=best,
                one_thread=points[0], all_threads=points[-1])


# ---- precision of the benchmarked modes against the float64 oracle --------------------------------------------------
MODES = {"mixed": (torch.float32, torch.bfloat16), "bf16": (torch.bfloat16, torch.bfloat16), "fp32": (torch.float32, torch.float32)}


def precision_vs_oracle(workload, modes, device, graphs=None):
    """Fused path in each mode vs the float64 oracle on a sample at the real dims, dropout 0 (train-mode BatchNorm):
    relative loss error and the relative L2 error of every parameter gradient (worst / median over the tensors whose
    exact gradient is not ~0).  For the reduced-precision modes each tensor's error is also divided by the ORACLE's own
    response of that tensor to bf16-sized (2^-9 relative) perturbations of the GEMM weights the mode rounds
    (oracle/noise.py): `worst_vs_oracle_noise` is the largest such ratio -- an ill-conditioned gradient (GINConv.eps) has a
    large error AND a large noise floor, a wrong kernel a large ratio.  Since r4 the perturbation covers the activations
    and activation gradients the mode rounds as well (reference_math's storage taps).  Same metric as tests/test_hip_configs.py,
    which bounds the ratio by 4 ON THIS SAMPLE too (measured worst: 2.0; the noise is the maximum over four perturbation seeds since r5 --
    over two it under-estimated a heavy-tailed response: GINConv.eps read 7 x)."""
    from graphtrans_amd import ops as gt_ops
    from oracle import noise as on
    from oracle import reference_math as rm
    if workload not in ("code2", "molpcba", "er"):
        return None
    graphs = graphs or {"er": 4}.get(workload, 24)
    out = {"sample": f"{graphs} graphs, dropout 0, vs oracle/reference_math.py in float64; vs_oracle_noise = rel-L2 error / the "
                     "oracle's own rel-L2 response to 2^-9 relative perturbations of the GEMM weights, activations and activation gradients the mode rounds to bf16"}
    ref = None
    for mode in modes:
        matmul, tok = MODES[mode]
        gt_ops.set_matmul_dtype(matmul)
        torch.manual_seed(7)
        args, model, gen, loss_fn, _ = build(workload, tok, "cpu", graphs)
        args.gnn_dropout = args.transformer_dropout = 0.0
        model.gnn_node.drop_ratio = 0.0
        model.transformer_encoder.dropout_p = 0.0
        b = gen(5)
        oargs = on.oracle_args(args)
        loss_of = {"code2": lambda o_: rm.code2_loss(o_, b.y_arr), "molpcba": lambda o_: rm.mol_loss(o_, b.y),
                   "er": lambda o_: rm.tud_loss(o_, b.y)}[workload]
        if ref is None:   # identical parameters in every mode (same seed): one oracle run
            sd = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
            b64 = b
            if getattr(b, "x", None) is not None and b.x.is_floating_point():
                import copy
                b64 = copy.copy(b)
                b64.x = b.x.double()
                if getattr(b, "edge_attr", None) is not None and b.edge_attr.is_floating_point():
                    b64.edge_attr = b.edge_attr.double()
            torch.set_default_dtype(torch.float64)
            try:
                l64 = loss_of(rm.gnn_transformer(sd, oargs, b64, None, True))
                l64.backward()
            finally:
                torch.set_default_dtype(torch.float32)
            ref = (float(l64.detach()), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}, sd, b64)
        model = model.to(device).train()
        bd = attach_sizes(b).to(device)
        loss = loss_fn(model(bd), bd)
        loss.backward()
        torch.cuda.synchronize()
        l64v, g64, sd64, b64 = ref
        rms = {k: float(r.norm()) / r.numel() ** 0.5 for k, r in g64.items()}
        top = max(rms.values())
        errs = {}
        for k, p in model.named_parameters():
            if k in g64 and rms[k] >= 1e-3 * top:
                errs[k] = float((p.grad.detach().double().cpu() - g64[k]).norm() / g64[k].norm())
        worst = max(errs, key=errs.get)
        vals = sorted(errs.values())
        out[mode] = dict(loss_rel_err=float(f"{abs(float(loss) - l64v) / abs(l64v):.3e}"), grad_rel_l2_worst=float(f"{errs[worst]:.3e}"),
                         grad_rel_l2_worst_param=worst, grad_rel_l2_median=float(f"{vals[len(vals) // 2]:.3e}"), tensors=len(errs))
        if mode != "fp32":
            floor = on.lowp_noise({k: v.detach() for k, v in sd64.items()}, oargs, b64, rm.gnn_transformer, loss_of, g64, mode)
            ratio = {k: errs[k] / max(floor[k], 2.5e-4) for k in errs}
            wr = max(ratio, key=ratio.get)
            out[mode].update(worst_vs_oracle_noise=float(f"{ratio[wr]:.3g}"), worst_vs_oracle_noise_param=wr,
                             oracle_noise_of_worst_err_param=float(f"{floor[worst]:.3e}"),
                             median_vs_oracle_noise=float(f"{sorted(ratio.values())[len(ratio) // 2]:.3g}"))
        del model
    return out


# ---- one timed measurement ------------------------------------------------------------------------------------------
def measure(opt, mode, scaling, world, rank, device, want_kernels=True):
    """Build the model of `mode`, run opt.warmup + opt.steps steps, return (result dict for rank 0, model, args)."""
    import gc
    try:
        return _measure(opt, mode, scaling, world, rank, device, want_kernels)
    finally:
        gc.unfreeze()   # _measure freezes the set-up heap after its warm-up: undone on every way out (a raise included)


def _measure(opt, mode, scaling, world, rank, device, want_kernels):
    import torch.distributed as dist

    from graphtrans_amd import _lib
    from graphtrans_amd import ops as gt_ops
    from graphtrans_amd.dist import GradSync
    from graphtrans_amd.optim import FusedAdamW

    matmul_dtype, dtype = MODES[mode]   # (GNN-side GEMM compute, token rows)
    gt_ops.set_matmul_dtype(matmul_dtype)
    per_gpu = opt.batch or {"nci1": 32, "code2-pna": 128}.get(opt.workload, 256)
    if scaling == "strong":
        if per_gpu % world:
            raise SystemExit("--scaling strong: the global batch %d does not divide over %d GPUs" % (per_gpu, world))
        per_gpu //= world
    torch.manual_seed(1234)  # identical initial parameters on every rank
    args, model, gen, loss_fn, wl_name = build(opt.workload, dtype, device, per_gpu)
    model.train()
    # strong scaling splits ONE global batch over the ranks: with per-rank BatchNorm statistics that is a different model from the
    # reference's single-device b256 (modules/gnn_module.py:204) -- synchronised statistics restore it, on the fused path
    # (dist.BnSyncHook: 2 D + 1 floats per BatchNorm and direction over RCCL)
    sync_bn = world > 1 and scaling == "strong" and not opt.no_sync_bn
    if sync_bn:
        from graphtrans_amd.modules.norm import convert_sync_batchnorm
        convert_sync_batchnorm(model)
    sync = GradSync(model.parameters(), world_size=world).attach(model)
    optim = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)  # torch.optim.AdamW semantics, one HIP launch
    torch.manual_seed(1234 + rank)  # per-rank dropout streams
    if scaling == "strong" and opt.workload in RAW_GEN and world > 1:
        # one GLOBAL batch per seed (identical on every rank), dealt to the ranks by size so that the quadratic
        # attention cost is even (dist.balanced_shards), assembled on the device from the raw graphs
        from graphtrans_amd.dist import balanced_shards
        batches = []
        for i in range(4):
            st_i = make_store(opt.workload, per_gpu * world, i)
            batches.append(st_i.collate(balanced_shards(st_i.nodes, world)[rank]))
    else:
        batches = [attach_sizes(gen(1000 * rank + i)).to(device) for i in range(4)]  # .to() keeps the host-side sizes
    store = None
    if opt.from_store:
        store = make_store(opt.workload, 4 * per_gpu, 1000 * rank)
        sampler = np.random.default_rng(rank)

    def step(i):
        if store is not None:
            b = store.collate(sampler.permutation(len(store))[:per_gpu])
        else:
            b = batches[i % len(batches)]
            b.__dict__.pop("_gt_structure", None)  # graph_prep is part of the step
        sync.zero()
        out = model(b)
        loss = loss_fn(out, b)
        loss.backward()
        sync.finish()
        if not opt.no_optimizer:
            optim.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(opt.warmup):
        step(i)
    barrier()
    # The interpreter's cyclic collector: a step allocates a few hundred tracked objects, so a full (generation-2) collection comes
    # round every ~100 steps and walks the WHOLE heap -- ~1 M objects of torch / sympy module state, 50-80 ms a pass = 0.4 ms per
    # step amortised on a 0.9 ms step (tools/host_phases.py with GT_GC=on|off|freeze: NCI1 1.28 / 0.91 / 0.88 ms).  gc.freeze() moves
    # what exists after set-up into the permanent generation; collections still run, over the step's own objects only.
    import gc
    gc.collect()
    gc.freeze()
    # HIP events around the aggregate / attention / linear launches of every 24th timed step (each event pair
    # costs ~3 us of stream time and the dW GEMMs stay on the main stream while bracketed, i.e. nothing overlaps them: a
    # sampled step is ~30 % slower; sampling keeps the timed region within ~2 % of a run with --no-kernel-timing)
    timing = want_kernels and not opt.no_kernel_timing
    off = 8 if opt.steps > 8 else opt.steps // 2   # short runs still get one sampled step
    sample = (lambda i: i % 24 == off) if timing else (lambda i: False)
    L = _lib.lib()
    if timing:
        _lib.profile_enable(PROF_MASK)
        L.gt_profile_enable(0)  # pool allocated, records cleared; recording toggled per step below
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(opt.steps + 1)]   # per-step device times (median)

    cur = torch.cuda.current_stream(device.index)   # (without an index torch resolves the device through hipGetDeviceCount: 110 us per call)
    t0 = time.perf_counter()
    for i in range(opt.steps):
        marks[i].record(cur)
        if sample(i):
            L.gt_profile_resume(PROF_MASK)
        loss = step(opt.warmup + i)
        if sample(i):
            L.gt_profile_resume(0)
    marks[opt.steps].record(cur)
    t_enqueued = time.perf_counter() - t0  # host time to enqueue all steps (no sync inside)
    barrier()
    elapsed = time.perf_counter() - t0
    records = _lib.profile_records() if timing else []
    _lib.profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())
    # the host's own cost of a step: enqueue time measured from an IDLE device, one step at a time (in the timed loop above a
    # GPU-bound step makes the host wait for queue space, so host_enqueue_ms_per_step there is bounded below by the device time)
    idle = []
    for i in range(min(12, max(opt.steps, 1))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(opt.warmup + opt.steps + i)
        idle.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    idle.sort()
    host_idle_ms = 1e3 * idle[len(idle) // 2]
    dp_stats = dict(sync.stats)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(opt.steps) if not sample(i))
    res = None
    if rank == 0:
        total_graphs = per_gpu * world * opt.steps
        nodes = int(np.mean([b.num_nodes for b in batches]))
        edges = int(np.mean([b.edge_index.shape[1] for b in batches]))
        res = {
            "value": round(total_graphs / elapsed, 1), "ms_per_step": round(1e3 * elapsed / opt.steps, 4),
            "ms_per_step_median_device": round(step_ms[len(step_ms) // 2], 4) if step_ms else None,
            "scaling": scaling,
            # arithmetic of the GEMMs: message-passing side + heads / encoder layers (accumulators, statistics, softmax are fp32)
            "dtype": {"mixed": "fp32+bf16", "bf16": "bf16", "fp32": "fp32"}[mode],
            "config": {"workload": wl_name, "mode": mode, "graphs_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "avg_nodes_per_batch": nodes, "avg_edges_per_batch": edges,
                       "parallelism": f"dp{world} (graph-sharded, RCCL grad all-reduce {sync.grad_bytes() >> 20} MiB)",
                       "step": ("collate+" if store is not None else "") + "zero_grad+graph_prep+fwd+loss+bwd+allreduce" + ("" if opt.no_optimizer else "+AdamW"),
                       "gnn_dtype": "fp32 storage, %s (message passing, VN MLP, gnn2transformer, heads)" % (
                           "bf16 MFMA GEMMs" if matmul_dtype == torch.bfloat16 else
                           ("fp32-accurate GEMMs: bf16x6 on the bf16 matrix pipe for the big-M linears (three-way bf16 split of both operands, six products, "
                            "fp32 accumulation), exact-fp32 MFMA for the short-M ones" if _w3_enabled() else "exact-fp32 MFMA GEMMs")),
                       "transformer_dtype": "%s token rows, %s (encoder layers)" % (
                           ("bf16", "bf16 MFMA") if dtype == torch.bfloat16 else
                           ("fp32", "bf16 MFMA" if matmul_dtype == torch.bfloat16 else
                            ("fp32-accurate products on the bf16 pipe (bf16x6: GEMMs and attention)" if _w3_enabled() else "exact-fp32 MFMA"))),
                       "dropout": {"gnn": args.gnn_dropout, "transformer": args.transformer_dropout},
                       "batchnorm": "synchronised over the ranks (statistics of the global batch)" if sync_bn else "per-rank statistics",
                       "interpreter": "gc.freeze() after warm-up (set-up heap out of the cyclic collector's full passes)"},
            "final_loss": round(final_loss, 5), "host_enqueue_ms_per_step": round(1e3 * t_enqueued / opt.steps, 3),
            "host_enqueue_ms_per_step_idle_device": round(host_idle_ms, 3),
        }
        if world > 1:
            import torch.distributed as dist
            nsteps = opt.warmup + opt.steps + len(idle)
            hook = None
            try:
                from graphtrans_amd import engine as _eng
                hook = _eng.state(model).get("bn_hook")
            except Exception:
                pass
            res["data_parallel"] = {
                "backend": dist.get_backend(), "ranks": dist.get_world_size(), "rank0_device": str(device),
                "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                "grad_allreduce_collectives_per_step": round(dp_stats["collectives"] / max(nsteps, 1), 2),
                "grad_allreduce_mb_per_step": round(dp_stats["bytes"] / max(nsteps, 1) / 1e6, 2),
                "syncbn_exchanges_per_step": round(hook.calls / max(nsteps, 1), 1) if hook is not None else 0,
            }
        if records:
            d_model, max_len = args.d_model, int(args.max_input_len)
            fl = []
            for b in batches:  # attention core flops over VALID lengths: 4 n^2 d per graph and layer
                n = np.minimum(np.asarray(b._sizes, dtype=np.float64), max_len) + (1 if args.graph_pooling == "cls" else 0)
                fl.append(float((4.0 * n * n * d_model).sum()))
            rep = kernel_report(records, float(np.mean(fl)), dtype, matmul_dtype)
            # "traffic": HBM-side bytes per call from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3
            # --pmc passes over this very command: tools/pmc_round.sh), attached only when measured on THIS build
            tr, note = pmc_traffic(opt.workload, mode, per_gpu)
            for k in rep:
                if k in tr:
                    rep[k]["traffic"] = tr[k]
                elif k.split("[")[0] in tr and sum(1 for q in rep if q.split("[")[0] == k.split("[")[0]) == 1:
                    rep[k]["traffic"] = tr[k.split("[")[0]]
            if rep:
                fam = kernel_report(records, float(np.mean(fl)), dtype, matmul_dtype, pool=True)
                dom = max(fam, key=lambda k: fam[k]["total_ms"])
                r = dict(fam[dom])
                r["kernel"] = dom
                def base_tag(k):   # "k_lin1[dx][bf16]" -> ("k_lin1[dx]", "[bf16]"): the dtype / rows tag kernel_report appends
                    for t in ("[bf16", "[fp32"):
                        if t in k:
                            return k[:k.index(t)], k[k.index(t):]
                    return k, ""
                members = [k for k in rep if k != dom and pooled_name(base_tag(k)[0]) + base_tag(k)[1] == dom]
                if dom in rep:
                    r["traffic"] = rep[dom]["traffic"]
                elif members:   # a pooled family: call-weighted mean of its members' PMC bytes (or the pooled PMC row itself)
                    r["members"] = members
                    pooled_key = base_tag(dom)[0] + "[fwd|dx]" + base_tag(dom)[1]
                    tv = [(rep[m]["traffic"], rep[m]["calls"]) for m in members]
                    if dom in tr or pooled_key in tr:
                        r["traffic"] = tr.get(dom, tr.get(pooled_key))
                    elif all(t is not None for t, _ in tv):
                        r["traffic"] = int(sum(t * c for t, c in tv) / sum(c for _, c in tv))
                r["timing"] = "HIP events on the launch stream, every 24th timed step"
                r["traffic_source"] = note
                res["roofline"] = r
                res["kernels"] = rep
    del optim, sync
    return res, model, args, per_gpu


LINE_LIMIT = 4096   # the driver keeps the tail of stdout: the one JSON line must fit with room to spare (21 KB -> parsed: null)
REPORT_NAME = "bench_report.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def compact_line(res, report_path=None):
    """The ONE line rank 0 prints: the contract keys, `config` (workload + mode + sizes, no prose), `roofline` of the dominant
    kernel, `cpu_baseline` as numbers, the fp32-contract / bf16 values, three precision numbers per mode, the other scaling for
    N > 1 and the path of the full report.  Everything else of `res` lives in bench_report.json.  Never longer than LINE_LIMIT
    bytes: optional blocks are dropped from the back until it fits (tests/test_bench_line.py)."""
    line = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = res.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "mode", "graphs_per_gpu", "global_batch", "avg_nodes_per_batch", "avg_edges_per_batch", "parallelism", "step"))
    if isinstance(line["config"].get("parallelism"), str):
        line["config"]["parallelism"] = line["config"]["parallelism"].split(" ")[0]
    rf = res.get("roofline")
    if rf:
        line["roofline"] = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_us", "calls", "algorithmic_flops", "algorithmic_bytes"))
        line["roofline"]["traffic"] = rf.get("traffic")
        src = str(rf.get("traffic_source", ""))
        line["roofline"]["traffic_source"] = src if len(src) <= 120 else src[:117] + "..."
    cb = res.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "s_per_step"))
        c["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:160]
        if cb.get("full_batch"):
            c["full_batch_value"] = cb["full_batch"].get("value")
        if cb.get("one_thread"):
            c["one_thread_value"] = cb["one_thread"].get("value")
        line["cpu_baseline"] = c
    optional = []   # (key, value) in the order they are dropped LAST .. FIRST when the line would not fit
    modes = dict(res.get("modes") or {})
    if cfg.get("mode") in ("mixed", "bf16", "fp32"):
        modes[cfg["mode"]] = {"value": res.get("value"), "ms_per_step": res.get("ms_per_step")}
    if "fp32" in modes:
        line["value_fp32_contract"] = modes["fp32"].get("value")
        line["ms_per_step_fp32_contract"] = modes["fp32"].get("ms_per_step")
    if "bf16" in modes:
        line["value_bf16"] = modes["bf16"].get("value")
    if "mixed" in modes and cfg.get("mode") != "mixed":
        line["value_mixed"] = modes["mixed"].get("value")
    pv = res.get("precision_vs_oracle")
    if isinstance(pv, dict):
        optional.append(("precision_vs_oracle", {m: _pick(v, ("loss_rel_err", "grad_rel_l2_worst", "grad_rel_l2_median"))
                                                 for m, v in pv.items() if isinstance(v, dict)} if "error" not in pv else {"error": str(pv["error"])[:200]}))
    for k in ("strong_scaling", "weak_scaling"):
        if k in res:
            optional.append((k, _pick(res[k], ("value", "ms_per_step", "graphs_per_gpu", "global_batch", "error")) |
                             ({"per_rank_batchnorm_value": res[k]["per_rank_batchnorm"].get("value")} if isinstance(res[k].get("per_rank_batchnorm"), dict) else {})))
    if "data_parallel" in res:
        optional.append(("data_parallel", _pick(res["data_parallel"], ("backend", "ranks", "rccl_ranks", "grad_allreduce_mb_per_step", "grad_allreduce_collectives_per_step"))))
    ag = res.get("aggregate_stress")
    if isinstance(ag, dict) and "gt_aggregate_fwd" in ag:
        optional.append(("aggregate_stress", {"N": ag.get("N"), "E": ag.get("E"), "D": ag.get("D"),
                                              "fwd_frac_hbm": ag["gt_aggregate_fwd"].get("frac"), "fwd_us": ag["gt_aggregate_fwd"].get("avg_us"),
                                              "bwd_frac_hbm": (ag.get("gt_aggregate_bwd") or {}).get("frac"), "bwd_us": (ag.get("gt_aggregate_bwd") or {}).get("avg_us")}))
    optional.append(("host_enqueue_ms_per_step_idle_device", res.get("host_enqueue_ms_per_step_idle_device")))
    optional.append(("launches_per_step", res.get("launches_per_step")))
    if report_path:
        line["report"] = report_path
    for k, v in optional:
        if v is not None:
            line[k] = v
    text = json.dumps(line)
    while len(text) > LINE_LIMIT and optional:   # never reached on real records; the contract keys, roofline and cpu_baseline always stay
        k, _ = optional.pop()
        line.pop(k, None)
        text = json.dumps(line)
    if len(text) > LINE_LIMIT:
        raise RuntimeError("bench line of %d bytes even without the optional blocks" % len(text))
    return text


def emit(res, path=None):
    """bench_report.json (the full record; --report PATH) next to this script, then the compact line as the LAST line of stdout"""
    shown = path or REPORT_NAME
    path = path or os.path.join(REPO, REPORT_NAME)
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
    except OSError as e:
        shown = "not written: %r" % (e,)
    sys.stdout.flush()
    print(compact_line(res, shown), flush=True)


def self_spawn(opt):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(opt.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="code2", choices=["code2", "molpcba", "nci1", "er", "code2-pna"])
    ap.add_argument("--batch", type=int, default=None, help="graphs per GPU (default 256; nci1 32)")
    ap.add_argument("--mode", default="mixed", choices=["mixed", "bf16", "fp32"], help="precision mode of the headline value (see the module docstring)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"], help="deprecated alias of --mode bf16|fp32")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (the headline): --batch graphs on EVERY GPU; strong: --batch is the global batch, split evenly "
                         "over the GPUs (b256 -> 32 graphs per GPU at 8).  With N > 1 the other one is measured too and "
                         'reported as "strong_scaling" / "weak_scaling" in the same JSON line')
    ap.add_argument("--no-extra", action="store_true", help="only the headline measurement (no other modes / scaling / precision report)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="BASELINE.md section 3 in full (minutes of CPU time) instead of the bounded sample")
    ap.add_argument("--no-sync-bn", action="store_true", help="strong scaling with per-rank BatchNorm statistics (default: synchronised)")
    ap.add_argument("--no-optimizer", action="store_true", help="time zero+fwd+loss+bwd(+allreduce) only")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--report", default=None, help="where the full record goes (default: bench_report.json next to this script)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check only (runs without a GPU): rendezvous, barrier, max-over-ranks reduction and the "
                         "JSON line with value 0 -- what tests/test_dist_gloo.py uses to cover `--gpus N` self-spawn on gloo")
    ap.add_argument("--from-store", action="store_true",
                    help="assemble every batch on the device from an HBM graph store inside the timed step "
                         "(gt_collate: sampling + augment_edge + collation; code2 / molpcba / code2-pna)")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B only: a named runtime option of the library (include/graphtrans_hip.h, gt_option_set), e.g. lin_ring=1")
    opt = ap.parse_args()
    if opt.dtype is not None:
        opt.mode = opt.dtype
    for kv in opt.lib_option:
        from graphtrans_amd import _lib
        name, _, val = kv.partition("=")
        _lib.option_set(name, int(val))

    if "WORLD_SIZE" not in os.environ and opt.gpus > 1:
        self_spawn(opt)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if opt.dry_run:
        import torch.distributed as dist
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "value": 0.0, "unit": "graphs/s", "n_gpus": world, "steps": opt.steps,
                              "warmup": opt.warmup, "max_over_ranks": float(t.item()), "scaling": opt.scaling}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hooks (a 1-GPU box cannot host two RCCL ranks): GT_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # GT_BENCH_BACKEND=gloo reduces through the host, which exercises the whole multi-rank path of this script
    if os.environ.get("GT_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GT_BENCH_BACKEND", "nccl")
        # a rank that fails inside the second (other-scaling) measurement leaves its peers inside a collective: the group's timeout
        # turns that hang into an exception on them, which the try/except around that measurement records next to the headline
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("GT_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    if opt.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {opt.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    head, model, args, per_gpu = measure(opt, opt.mode, opt.scaling, world, rank, device)
    res = None
    if rank == 0:
        metric = "graphs/sec (fwd+bwd) OGBG-Code2 GCN-Virtual b256" if opt.workload == "code2" else f"graphs/sec (fwd+bwd) {opt.workload}"
        if world > 1:   # which of the two scalings `value` is (the other one rides in the same line)
            metric += f" [value: {opt.scaling} scaling, {head['config']['graphs_per_gpu']} graphs per GPU x {world} GPUs]"
        res = {"metric": metric,
               "value": head["value"], "unit": "graphs/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
               "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": opt.scaling, "vs_baseline": None,
               "dtype": head["dtype"], "data": "synthetic"}
        res.update({k: v for k, v in head.items() if k not in res})
    extra = not opt.no_extra
    if extra and world > 1:   # the other scaling of the same step, same ranks
        other = "strong" if opt.scaling == "weak" else "weak"
        del model
        try:   # the second measurement must not take the headline line down with it (every rank takes the same branch)
            o, model, _, _ = measure(opt, opt.mode, other, world, rank, device, want_kernels=False)
            if rank == 0:
                res[other + "_scaling"] = {k: o[k] for k in ("value", "ms_per_step", "scaling", "host_enqueue_ms_per_step", "host_enqueue_ms_per_step_idle_device", "data_parallel") if k in o} | \
                    {"graphs_per_gpu": o["config"]["graphs_per_gpu"], "global_batch": o["config"]["global_batch"],
                     "batchnorm": o["config"]["batchnorm"]}
            if other == "strong" and not opt.no_sync_bn:   # and what the synchronised statistics cost: the same split with per-rank statistics
                del model
                opt.no_sync_bn = True
                o2, model, _, _ = measure(opt, opt.mode, other, world, rank, device, want_kernels=False)
                opt.no_sync_bn = False
                if rank == 0:
                    res[other + "_scaling"]["per_rank_batchnorm"] = {k: o2[k] for k in ("value", "ms_per_step")}
        except Exception as e:
            model = None
            if rank == 0:
                res[other + "_scaling"] = {"error": repr(e)[:500]}
            extra = False   # (the group may be broken: nothing collective after this)
    if extra and world == 1:
        others = [m for m in ("mixed", "bf16", "fp32") if m != opt.mode]
        res["modes"] = {}
        for m in others:
            del model
            o, model, _, _ = measure(opt, m, opt.scaling, world, rank, device)
            res["modes"][m] = {k: o[k] for k in ("value", "ms_per_step", "ms_per_step_median_device", "dtype", "host_enqueue_ms_per_step", "host_enqueue_ms_per_step_idle_device", "final_loss") if k in o}
            res["modes"][m]["config"] = {k: o["config"][k] for k in ("mode", "gnn_dtype", "transformer_dtype")}
            if "roofline" in o:
                res["modes"][m]["roofline"] = o["roofline"]
                res["modes"][m]["kernels"] = o["kernels"]
        try:
            res["precision_vs_oracle"] = precision_vs_oracle(opt.workload, [opt.mode] + others, device)
        except Exception as e:   # the report must not take the bench line down
            res["precision_vs_oracle"] = {"error": repr(e)}
        MODES_RESET = MODES[opt.mode]
        from graphtrans_amd import ops as gt_ops
        gt_ops.set_matmul_dtype(MODES_RESET[0])
        try:
            res["aggregate_stress"] = aggregate_stress_report(device)
        except Exception as e:
            res["aggregate_stress"] = {"error": repr(e)}
    if rank == 0:
        if opt.workload in RAW_GEN:
            res["collate"] = collate_report(opt.workload, per_gpu, with_cpu=(world == 1 and not opt.no_cpu_baseline))
        if world == 1 and not opt.no_cpu_baseline:
            if opt.cpu_baseline_full:
                full = cpu_baseline_full(opt.workload, model, args)
                b = full["best"]
                res["cpu_baseline"] = dict(value=b["graphs_per_s"], unit="graphs/s", cores=b["threads"], kind="port",
                                           s_per_step=b["s_per_step"], one_thread={"value": full["one_thread"]["graphs_per_s"]},
                                           sample=f"oracle/reference_math.py fwd+loss+bwd fp32, full {full['graphs']}-graph batch, median of "
                                                  f"{b['timed_iterations']} after 2 warm-ups, best of the thread counts below", full=full)
            else:
                res["cpu_baseline"] = cpu_baseline(opt.workload, model, args, full_graphs=per_gpu)
        emit(res, opt.report)
    if world > 1:
        import torch.distributed as dist
        try:
            dist.barrier()   # rank 0 is still writing its report: leave the group together
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
