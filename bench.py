"""bench.py - NGCF 3-layer forward on MI355X: propagated edges/s + roofline of the SpMM kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2..3]): synthetic bipartite graph, 1 M users x 100 K items,
50 M drawn interactions (popularity-skewed items, de-duplicated; the actual count is reported), both triangles
stored -> nnz(L) = 2 x interactions; d0 = d = 128, 3 layers, fp32, random-init weights, batch of 1024 triplets.
Secondary fields of the same line at N = 1 (labelled, never the headline): the reference-legal width 130 through the whole
NGCF.forward, and the Seoul-shaped BASELINE configs 0-1 (c1, c2) replayed as hipGraphs.
One "step" = one pass of the hot path: E0 -> 3 x (L.E SpMM + fused dense/LeakyReLU/normalise) -> all_E,
3 row gathers, fused BPR loss  (NGCF.py:120-156 + bprloss.py:15-22).  The feature injection (NGCF.py:103-115)
is not part of the step at this width: the reference itself raises for embed_size = 128 (not a multiple of 5).
Unit of work: one stored nonzero of L processed in one layer; value = n_layers * nnz(L) * steps / time.
N > 1: the SAME graph is row-partitioned over the ranks (strong scaling), exchange scheme `--exchange`, rows moved by the
CU-free p2p exchange (copy engines; NGCF_DIST_COLLECTIVES=torch: RCCL collectives) - DESIGN.md 6.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth of the 8 XCDs (same guide, "L2 (per XCD)")

WORKLOADS = {
    # name: (n_user, n_item, interactions, d0, layers, seed)
    "c3": (1_000_000, 100_000, 50_000_000, 128, (128, 128, 128), 2603),
    "c5": (10_000_000, 1_000_000, 500_000_000, 256, (256, 256, 256), 2605),
    # C3's graph at the width the reference can actually run (embed_size a multiple of 5): whole model.forward()
    "c3_130": (1_000_000, 100_000, 50_000_000, 130, (128, 128, 128), 2603),
    "small": (100_000, 10_000, 2_000_000, 128, (128, 128, 128), 2603),
    # BASELINE.json configs[0..1]: Seoul-shaped stand-in graph, full nn.Module forward incl. injection
    "c1": (5840, 100, 0, 65, (64, 64), 1801),
    "c2": (5840, 100, 0, 515, (512, 512), 1801),
    # the reference's own TRAINING configuration (main.py:63-76, parsers.py defaults, experiment.py:45-58) on the Seoul-shaped
    # stand-in: embed 65 -> [65, 65, 65], node dropout 0.3, message dropout [0.1]*3, batch 1024, Adam lr 1e-3, BPR wd 0.025,
    # node_flag=True, model.train(); one step = forward + BPR + backward + optimizer step
    "c1_train": (5840, 100, 0, 65, (65, 65, 65), 1801),
    # the TRAINING step at scale (experiment.py:45-58 on C3's graph): embed 130 -> [128]*3 (the reference-legal width next to d = 128),
    # node dropout 0.3 and message dropout 0.1 drawn on the device, batch 1024, Adam; step = forward + BPR + backward + optimizer step
    "c3_train": (1_000_000, 100_000, 50_000_000, 130, (128, 128, 128), 2603),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--exchange", default="bipartite", choices=["bipartite", "allgather"],
                    help="N > 1: the scheme `value` is measured on.  bipartite (default): row partition of the users, item partial "
                         "sums reduce-scattered to their owners, all-gather of the owned items' carry rows per layer; allgather: the "
                         "literal scheme of BASELINE config 4 (all-gather of every rank's user and item carry rows per layer, 10x the "
                         "bytes).  The other one is timed as a labelled secondary field of the same line")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (other exchange "
                    "scheme at N > 1; the reference-legal 130-wide first layer through model.forward() at N = 1)")
    ap.add_argument("--seg-len", type=int, default=0, help="override the row-segment length (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--uniform-items", action="store_true", help="secondary line: no popularity skew")
    ap.add_argument("--hipgraph", action="store_true", help="Seoul-sized workloads: replay the forward as a hipGraph")
    ap.add_argument("--dropout-mode", default="reference", choices=["reference", "device"],
                    help="c1_train: where the dropout masks come from (NGCF.node_dropout_mode / mess_dropout_mode); the other "
                         "mode is timed as a labelled secondary field")
    return ap.parse_args()


def host_cores():
    """Host threads for the CPU baseline: this process's share of the node (a 1-GPU box gets 16 cores; the affinity
    mask may list every CPU of the node, and hundreds of threads only thrash on these matrix sizes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 16))


def spmm_model_a_bytes(nnz, n_rows, n_cols, d):
    """Compulsory traffic of one SpMM launch (SURVEY.md 8d model A): CSR once, E read once, LE written once."""
    return nnz * 8 + (n_rows + 1) * 8 + n_cols * d * 4 + n_rows * d * 4


def parity_spot_check(ref, model, coo, seed):
    """inside the same run, compare the engine's all_E with the CPU oracle's on 4 096 random rows plus the
    64 heaviest rows, over the columns the oracle sample covers, at the forward tolerance (atol 2e-5, rtol 2e-3)."""
    N = ref.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed + 7)
    deg = torch.bincount(coo["rows"].cpu(), minlength=N)
    pick = torch.cat([torch.randint(0, N, (min(4096, N),), generator=g), torch.topk(deg, min(64, N)).indices]).unique()
    got = torch.cat([model.all_users_emb, model.all_items_emb], 0)[pick.to(model.all_users_emb.device), :ref.shape[1]].cpu()
    want = ref[pick]
    err = (got - want).abs()
    ok = bool((err <= 2e-5 + 2e-3 * want.abs()).all())
    return {"rows": int(pick.numel()), "cols": int(ref.shape[1]), "max_abs_err": float(err.max()), "atol": 2e-5,
            "rtol": 2e-3, "ok": ok, "against": "oracle/ngcf_oracle.py propagate_torch (CPU restatement of NGCF.py:120-147)"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(coo, model, n_threads, seed=0):
    """The reference's PyTorch CPU path (oracle/ngcf_oracle.py, bit-exact restatement) on a bounded sample:
    ONE propagation layer (layer 1 of 3) of the same graph and weights, on this node's host cores.  Protocol of SURVEY
    8d: 1 warm-up + 3 timed repetitions, median, once with all of this process's host threads and once with 1."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ngcf_oracle as orc
    N = coo["n_user"] + coo["n_item"]
    idx = torch.stack([coo["rows"], coo["cols"]]).cpu()
    L = torch.sparse_coo_tensor(idx, coo["vals"].cpu(), (N, N))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}

    def one_layer():
        with torch.no_grad():
            return orc.propagate_torch(L, sd["user_embedding.weight"], sd["item_embedding.weight"], [sd["w1_list.0.weight"]],
                                       [sd["w1_list.0.bias"]], [sd["w2_list.0.weight"]], [sd["w2_list.0.bias"]])

    def median_of_3(threads):
        torch.set_num_threads(threads)
        ref, times = one_layer(), []                          # warm-up (its result is the parity reference)
        for _ in range(3):
            t0 = time.perf_counter()
            one_layer()
            times.append(time.perf_counter() - t0)
        return ref, sorted(times)[1], times

    ref, dt, times = median_of_3(n_threads)
    _, dt1, times1 = median_of_3(1)
    torch.set_num_threads(n_threads)
    return {"parity": parity_spot_check(ref, model, coo, seed), "value": coo["nnz"] / dt, "unit": "edges/s", "cores": n_threads,
            "kind": "port", "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"1 of 3 layers (SpMM + 3 Linear + LeakyReLU + normalize + cat) of the same graph, "
                      f"nnz(L)={coo['nnz']}, d={sd['user_embedding.weight'].shape[1]}, torch {torch.__version__} CPU; "
                      f"1 warm-up + 3 timed runs, median {dt:.1f} s at {n_threads} threads "
                      f"(16 of the node's host CPUs = a one-GPU box's share; the sparse mm is single-threaded in torch CPU, so more threads change "
                      f"nothing: 1 thread is within a few percent, see one_thread)",
            "seconds": dt, "runs_seconds": [round(t, 2) for t in times],
            "one_thread": {"value": coo["nnz"] / dt1, "unit": "edges/s", "cores": 1, "seconds": dt1,
                           "runs_seconds": [round(t, 2) for t in times1]}}


def cpu_baseline_full(coo, model, n_threads, seed=0):
    """Seoul-sized workloads: the whole propagation (all layers) of the CPU oracle, median of 5 runs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ngcf_oracle as orc
    N = coo["n_user"] + coo["n_item"]
    L = torch.sparse_coo_tensor(torch.stack([coo["rows"], coo["cols"]]).cpu(), coo["vals"].cpu(), (N, N))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    n_layer = model.n_layer
    w = [[sd[f"{nm}.{k}.{t}"] for k in range(n_layer)] for nm, t in (("w1_list", "weight"), ("w1_list", "bias"),
                                                                     ("w2_list", "weight"), ("w2_list", "bias"))]
    torch.set_num_threads(n_threads)
    times = []
    with torch.no_grad():
        for _ in range(6):
            t0 = time.perf_counter()
            ref = orc.propagate_torch(L, sd["user_embedding.weight"], sd["item_embedding.weight"], *w)
            times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[2]
    return {"parity": parity_spot_check(ref, model, coo, seed), "value": n_layer * coo["nnz"] / dt, "unit": "edges/s", "cores": n_threads, "kind": "port",
            "sample": f"whole {n_layer}-layer propagation, median of 5 after 1 warm-up, torch {torch.__version__} CPU, {dt * 1e3:.1f} ms",
            "seconds": dt}


TRAIN_CFG = dict(embed=65, layers=(65, 65, 65), node_dropout=0.3, mess_dropout=(0.1, 0.1, 0.1), lr=1e-3, wd=0.025)


def seoul_train_setup(pkg, dev, batch, mode, seed=1801, graphed=False, auto_graph=True, eval_mode=False):
    """The reference's training configuration (main.py:63-76 + parsers.py defaults) on the Seoul-shaped stand-in graph, and one
    step of experiment.py:45-58: model(node_flag=True) -> zero_grad -> BPR -> backward -> Adam.step, in train mode, with the
    module's defaults otherwise (index check on)."""
    slices = pkg.graphs.seoul_standin(dev, seed=seed)
    nu, ni = slices[0]["n_user"], slices[0]["n_item"]
    nd = {"user": nu, "item": ni, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(seed)
    c = TRAIN_CFG
    model = pkg.NGCF(c["embed"], list(c["layers"]), c["node_dropout"], list(c["mess_dropout"]), 1.0,
                     [pkg.graphs.to_sparse_coo(x) for x in slices], nd, batch, dev).to(dev)
    model.train(not eval_mode)       # eval_mode: the reference's loop from its 2nd epoch on (experiment.py:61,72: eval() is never undone)
    model.node_dropout_mode = model.mess_dropout_mode = mode
    model.auto_train_graph = bool(auto_graph)      # default on: in device mode forward and backward replay two captured graphs from the 2nd call on
    fused = os.environ.get("NGCF_BENCH_ADAM_FUSED") == "1"        # lab: torch's fused Adam (one launch) instead of the reference's default
    opt = torch.optim.Adam(model.parameters(), lr=c["lr"], capturable=bool(graphed), **({"fused": True} if fused else {}))
    crit = pkg.BPR(c["wd"], batch).to(dev)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    ids = {k: torch.randint(0, hi, (batch,), generator=g).to(dev)
           for k, hi in (("u_id", nu), ("pos_item", ni), ("neg_item", ni), ("age", 76), ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
    ids["year"] = torch.full((batch,), 18, device=dev)

    def step():
        u, p, n = model(node_flag=True, **ids)
        opt.zero_grad()
        loss = crit(u, p, n)
        loss.backward()
        opt.step()
        return loss
    if graphed:        # the same step captured once and replayed (opt-in: seoul_tourism_recommendation_ngcf_amd.GraphedTrainStep)
        gts = pkg.GraphedTrainStep(model, crit, opt, ids, node_flag=True)
        return model, (lambda: gts(**ids)), slices[0], ids
    return model, step, slices[0], ids


def time_train_steps(step, steps, warmup):
    """(ms per step wall clock with a sync on both sides, ms per step the host spent issuing it - i.e. without the final wait).
    r04: the median of up to five blocks (each at least 10 steps) with the garbage collector off inside a block - one host-side
    stall inside a single block of a launch-bound step is worth tenths of a millisecond per step."""
    import gc
    gc.collect()
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    n_blocks = max(1, min(5, steps // 10))
    per_block = steps // n_blocks
    was = gc.isenabled()
    gc.disable()
    wall, issue = [], []
    try:
        for _ in range(n_blocks):
            t0 = time.perf_counter()
            for _ in range(per_block):
                loss = step()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) / per_block * 1e3)
            issue.append(t_issue / per_block * 1e3)
    finally:
        if was:
            gc.enable()
    assert torch.isfinite(loss).item(), "non-finite loss"
    mid = sorted(range(n_blocks), key=lambda i: wall[i])[n_blocks // 2]
    return wall[mid], issue[mid], float(loss)


def cpu_train_baseline(coo_slices_cpu, n_user, model, ids, batch, n_threads, steps=5):
    """The reference's training step on the host (oracle/ngcf_oracle.py, the torch CPU ops of NGCF.py:102-156 + bprloss.py with
    torch autograd and torch.optim.Adam - what experiment.py:45-58 runs on a CPU device), same graph, same initial state."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ngcf_oracle as orc
    c = TRAIN_CFG
    torch.set_num_threads(n_threads)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if k.startswith(("w1_list", "w2_list", "item_emb", "user_emb"))}
    feats = {"age": sd["age_emb.weight"], "sex": sd["sex_emb.weight"], "month": sd["month_emb.weight"],
             "day": sd["day_emb.weight"], "dow": sd["dow_emb.weight"]}
    n = len(c["layers"])
    opt = torch.optim.Adam(list(leaves.values()), lr=c["lr"])
    ids = {k: v.cpu() for k, v in ids.items()}
    L = coo_slices_cpu
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.feature_inject_torch(leaves["user_embedding.weight"], feats, ids["u_id"], ids["age"], ids["sex"], ids["month"],
                                     ids["day"], ids["dow"], 1.0)
        all_E = orc.propagate_torch(L, leaves["user_embedding.weight"], leaves["item_embedding.weight"],
                                    [leaves[f"w1_list.{k}.weight"] for k in range(n)], [leaves[f"w1_list.{k}.bias"] for k in range(n)],
                                    [leaves[f"w2_list.{k}.weight"] for k in range(n)], [leaves[f"w2_list.{k}.bias"] for k in range(n)],
                                    mess_dropout=list(c["mess_dropout"]), training=True, node_dropout=c["node_dropout"], node_flag=True)
        u, p, ng = orc.gather_torch(all_E, n_user, ids["u_id"], ids["pos_item"], ids["neg_item"])
        opt.zero_grad()
        loss = orc.bpr_torch(u, p, ng, c["wd"], batch)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return dt, float(loss.detach())


def train_secondary(pkg, dev, batch, mode, steps=30, warmup=5, graphed=False, auto_graph=True, eval_mode=False):
    """One labelled measurement of the reference's training step in one dropout mode."""
    import gc
    gc.collect()
    torch.cuda.empty_cache()                # (the previous secondary's graphs and pools are gone before this one builds its own)
    model, step, coo, _ = seoul_train_setup(pkg, dev, batch, mode, graphed=graphed, auto_graph=auto_graph, eval_mode=eval_mode)
    ms, ms_issue, loss = time_train_steps(step, steps, warmup)
    n_layer = len(TRAIN_CFG["layers"])
    return {"ms_per_step": ms, "host_issue_ms_per_step": ms_issue, "value": n_layer * coo["nnz"] / (ms * 1e-3), "unit": "edges/s",
            "loss": loss, "steps": steps, "dropout_mode": mode, "hipgraph": bool(graphed),
            "model_mode": "eval (experiment.py:61,72: the loop's steady state from epoch 2 on; message dropout off, node dropout on)" if eval_mode else "train",
            "forward_backward_graphs": bool(auto_graph and mode == "device" and not graphed and len(model._train_graphs) > 0),
            "note": f"main.py:63-76 / experiment.py:45-58 on the Seoul-shaped stand-in: embed {TRAIN_CFG['embed']} -> {list(TRAIN_CFG['layers'])}, "
                    f"node dropout {TRAIN_CFG['node_dropout']}, message dropout {list(TRAIN_CFG['mess_dropout'])}, batch {batch}, Adam lr "
                    f"{TRAIN_CFG['lr']}, BPR wd {TRAIN_CFG['wd']}, node_flag=True, train mode; step = forward + BPR + backward + Adam.step; "
                    f"masks: {'torch CPU generator, drawn where the reference draws them' if mode == 'reference' else 'counter hash inside the kernels'}"}


def main_train(args, pkg, dev):
    """--workload c1_train: the reference's own training step (forward + BPR + backward + Adam) on the Seoul-shaped stand-in."""
    mode = args.dropout_mode
    model, step, coo, ids = seoul_train_setup(pkg, dev, args.batch, mode)
    ms, ms_issue, loss = time_train_steps(step, args.steps, args.warmup)
    n_layer = len(TRAIN_CFG["layers"])
    out = {
        "metric": "NGCF 3-layer forward: propagated edges/sec + achieved HBM GB/s, d=128",
        "value": n_layer * coo["nnz"] / (ms * 1e-3), "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "host_issue_ms_per_step": ms_issue, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c1_train: the reference's training step (main.py:63-76, experiment.py:45-58) on the Seoul-shaped stand-in, "
                               f"{coo['n_user']} users x {coo['n_item']} items, nnz(L)={coo['nnz']}, embed {TRAIN_CFG['embed']} -> "
                               f"{list(TRAIN_CFG['layers'])}, node dropout {TRAIN_CFG['node_dropout']} ({mode} mode), message dropout "
                               f"{list(TRAIN_CFG['mess_dropout'])}, batch={args.batch}, Adam lr {TRAIN_CFG['lr']}, BPR wd {TRAIN_CFG['wd']}; "
                               "step = forward + BPR + backward + optimizer step (NOT the headline metric's forward-only step)",
                   "n_user": coo["n_user"], "n_item": coo["n_item"], "nnz_L": coo["nnz"], "d": TRAIN_CFG["embed"], "n_layers": n_layer,
                   "batch": args.batch, "dropout_mode": mode, "parallelism": "single GPU"},
        "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                     "note": "launch-bound: the working set (E 1.5 MB, CSR 7 MB) is cache-resident and a step is ~100 launches of a few "
                             "microseconds; an HBM fraction is not meaningful here (SURVEY 8d)"},
        "loss": loss,
    }
    if not args.no_secondary:
        other = "device" if mode == "reference" else "reference"
        try:
            out["secondary"] = {f"dropout_mode_{other}": train_secondary(pkg, dev, args.batch, other, args.steps, args.warmup)}
        except Exception as exc:  # noqa: BLE001
            out["secondary"] = {f"dropout_mode_{other}": {"error": repr(exc)[:300]}}
        try:       # device masks, every launch issued by the host (auto_train_graph off)
            out["secondary"]["device_masks_no_graphs"] = train_secondary(pkg, dev, args.batch, "device", args.steps, args.warmup, auto_graph=False)
        except Exception as exc:  # noqa: BLE001
            out["secondary"]["device_masks_no_graphs"] = {"error": repr(exc)[:300]}
        try:       # the step with device masks captured once and replayed (opt-in GraphedTrainStep; Adam(capturable=True))
            out["secondary"]["device_masks_hipgraph"] = train_secondary(pkg, dev, args.batch, "device", max(args.steps, 100), 10, graphed=True)
        except Exception as exc:  # noqa: BLE001
            out["secondary"]["device_masks_hipgraph"] = {"error": repr(exc)[:300]}
    if not args.no_cpu_baseline:
        N = coo["n_user"] + coo["n_item"]
        L = torch.sparse_coo_tensor(torch.stack([coo["rows"], coo["cols"]]).cpu(), coo["vals"].cpu(), (N, N))
        torch.manual_seed(1801)
        fresh, _, _, _ = seoul_train_setup(pkg, dev, args.batch, mode)          # same initial state as the timed model had
        dt, closs = cpu_train_baseline(L, coo["n_user"], fresh, ids, args.batch, host_cores())
        out["cpu_baseline"] = {"value": n_layer * coo["nnz"] / dt, "unit": "edges/s", "cores": host_cores(), "kind": "port",
                               "cpu_model": cpu_model(), "seconds": dt, "loss": closs,
                               "sample": f"the same training step on the host: oracle/ngcf_oracle.py (torch CPU ops of NGCF.py:102-156 + "
                                         f"bprloss.py) with torch autograd and torch.optim.Adam, median of 5 steps after 1, {dt * 1e3:.1f} ms "
                                         f"per step at {host_cores()} threads, torch {torch.__version__}"}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def timed_blocks(fn, blocks=5, steps=40, warmup=20):
    """`blocks` x `steps` calls of fn, each block between two synchronisations: (median, min, max) ms per step over the blocks.
    One block of 200 steps turns a single host-side stall (a full Python garbage collection is tens of milliseconds in a process
    with torch loaded) into +0.15 ms per step; the median of five blocks does not, and the collector is off while a block runs."""
    import gc
    for _ in range(warmup):
        last = fn()
    torch.cuda.synchronize()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    per = []
    try:
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(steps):
                last = fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / steps * 1e3)
    finally:
        if was:
            gc.enable()
    return sorted(per)[len(per) // 2], min(per), max(per), per, last


C3_TRAIN_CFG = dict(node_dropout=0.3, mess_dropout=(0.1, 0.1, 0.1), lr=1e-3, wd=0.025)


def c3_train_measure(pkg, lib, dev, lap, n_user, n_item, nnz, batch, steps, warmup, seed=2603):
    """The training step at scale: experiment.py:45-58 on the C3 graph at embed 130 -> [128]*3, node dropout 0.3 + message dropout
    0.1 drawn on the device (the reference-mode masks are ~100 M CPU-generator draws per layer: minutes per step on any host),
    batch 1024, torch.optim.Adam over all parameters; step = forward + BPR + backward + optimizer step.  Returns a dict with the
    step time and the live mean duration of one SpMM product (forward L.E and backward L^T.dLE alike: L is symmetric)."""
    import ctypes as C
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _, _, _, d0, layers, _ = WORKLOADS["c3_train"]
    c = C3_TRAIN_CFG
    nd = {"user": n_user, "item": n_item, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(seed)
    model = pkg.NGCF(d0, list(layers), c["node_dropout"], list(c["mess_dropout"]), 1.0, [lap], nd, batch, dev).to(dev).train()
    model.check_indices = False
    model.node_dropout_mode = model.mess_dropout_mode = "device"
    fused = os.environ.get("NGCF_BENCH_ADAM_FUSED") == "1"        # lab: torch's fused Adam (one pass over the tables) instead of the reference's default
    opt = torch.optim.Adam(model.parameters(), lr=c["lr"], **({"fused": True} if fused else {}))
    crit = pkg.BPR(c["wd"], batch)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    ids = {k: torch.randint(0, hi, (batch,), generator=g).to(dev)
           for k, hi in (("u_id", n_user), ("pos_item", n_item), ("neg_item", n_item), ("age", 76), ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
    ids["year"] = torch.full((batch,), 18, device=dev)

    def step():
        u, p, n = model(node_flag=True, **ids)
        opt.zero_grad()
        loss = crit(u, p, n)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    lib.ngcf_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    n_l, tot = C.c_int64(), C.c_double()
    _lib.check(lib.ngcf_prof_collect(C.byref(n_l), C.byref(tot)))
    lib.ngcf_prof_enable(0)
    assert torch.isfinite(loss).item(), "non-finite loss"
    csr = model.laplacian_csr(0)
    N = n_user + n_item
    return {"ms_per_step": ms, "value": len(layers) * nnz / (ms * 1e-3), "unit": "edges/s", "loss": float(loss), "steps": steps,
            "spmm_products_timed": int(n_l.value), "spmm_products_per_step": n_l.value / max(steps, 1),
            "mean_spmm_product_ms": tot.value / max(n_l.value, 1),
            "spmm_model_a_bytes": {str(d): spmm_model_a_bytes(csr.nnz, N, N, d) for d in sorted({d0, *layers})},
            "peak_memory_GiB": torch.cuda.max_memory_allocated() / 2 ** 30, "adam_fused": fused, "model": model, "ids": ids, "d0": d0, "layers": list(layers),
            "note": f"experiment.py:45-58 at scale: embed {d0} -> {list(layers)}, node dropout {c['node_dropout']} + message dropout "
                    f"{list(c['mess_dropout'])} (device masks), batch {batch}, Adam lr {c['lr']}, BPR wd {c['wd']}, node_flag=True, train mode; "
                    "step = forward + BPR + backward + Adam.step; the products timed are the full SpMMs of the step (3 forward L.E + the "
                    "backward's L^T.dLE of the dense layers; the row-sparse last layer's backward product is a different kernel)"}


def cpu_c3_train_sample(lap_cpu, model, ids, n_user, batch, n_threads):
    """cpu_baseline of c3_train on a BOUNDED sample: ONE propagation layer (130 -> 128) of the same graph and initial weights on the
    host - forward, BPR on the [E0 | norm(E1)] rows, backward, Adam - through the torch CPU oracle (eval-mode semantics: the
    reference's CPU-drawn masks would add ~100 M generator draws); one run, no warm-up (about half a minute)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ngcf_oracle as orc
    torch.set_num_threads(n_threads)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    names = ["user_embedding.weight", "item_embedding.weight", "w1_list.0.weight", "w1_list.0.bias", "w2_list.0.weight", "w2_list.0.bias"]
    leaves = {k: sd[k].requires_grad_(True) for k in names}
    opt = torch.optim.Adam(list(leaves.values()), lr=C3_TRAIN_CFG["lr"])
    ids = {k: v.cpu() for k, v in ids.items()}
    t0 = time.perf_counter()
    all_E = orc.propagate_torch(lap_cpu, leaves[names[0]], leaves[names[1]], [leaves[names[2]]], [leaves[names[3]]], [leaves[names[4]]],
                                [leaves[names[5]]])
    u, p, ng = orc.gather_torch(all_E, n_user, ids["u_id"], ids["pos_item"], ids["neg_item"])
    opt.zero_grad()
    loss = orc.bpr_torch(u, p, ng, C3_TRAIN_CFG["wd"], batch)
    loss.backward()
    opt.step()
    return time.perf_counter() - t0, float(loss.detach())


def main_c3_train(args, pkg, lib, dev):
    """--workload c3_train: the training step at scale as its own line."""
    n_user, n_item, n_inter, d0, layers, seed = WORKLOADS["c3_train"]
    coo = pkg.graphs.synthetic_bipartite(n_user, n_item, n_inter, seed=seed, device=dev, item_skew=not args.uniform_items)
    lap = pkg.graphs.to_sparse_coo(coo)
    r = c3_train_measure(pkg, lib, dev, lap, n_user, n_item, coo["nnz"], args.batch, args.steps, args.warmup, seed)
    model, ids = r.pop("model"), r.pop("ids")
    a = r["spmm_model_a_bytes"]
    per_product = (a[str(d0)] + 2 * a[str(layers[0])]) / 3            # the forward's three products (130, 128, 128 wide); the backward's are 128 wide
    achieved = per_product / (r["mean_spmm_product_ms"] * 1e-3) / 1e9
    out = {
        "metric": "NGCF 3-layer forward: propagated edges/sec + achieved HBM GB/s, d=128",
        "value": r["value"], "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"c3_train: the TRAINING step (experiment.py:45-58) on synthetic bipartite {n_user} users x {n_item} items, "
                               f"{coo['interactions']} interactions, nnz(L)={coo['nnz']}; " + r["note"] +
                               " - NOT the headline metric's forward-only step: edges/s here = n_layers * nnz(L) / step time",
                   "n_user": n_user, "n_item": n_item, "interactions": coo["interactions"], "nnz_L": coo["nnz"], "d": d0,
                   "n_layers": len(layers), "batch": args.batch, "parallelism": "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "spmm_swept_kernel (one SpMM product of the step, forward L.E or backward L^T.dLE)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": per_product, "launches_timed": r["spmm_products_timed"],
                     "mean_launch_ms": r["mean_spmm_product_ms"],
                     "note": "model A per product (SURVEY 8d); the matrix-core kernels of the backward (bwd_weight_kernel, "
                             "layer_bwd_input_kernel) are priced in DESIGN.md 7 from profiles/r04_c3_train_kernel_stats.csv"},
        "train": {k: v for k, v in r.items() if k not in ("value", "unit")},
        "loss": r["loss"],
    }
    if not args.no_cpu_baseline:
        dt, closs = cpu_c3_train_sample(lap.cpu(), model, ids, n_user, args.batch, host_cores())
        out["cpu_baseline"] = {"value": coo["nnz"] / dt, "unit": "edges/s", "cores": host_cores(), "kind": "port", "cpu_model": cpu_model(),
                               "seconds": dt, "loss": closs,
                               "sample": f"ONE of the three layers ({d0} -> {layers[0]}) of the same graph and initial weights as a training step "
                                         f"on the host: oracle/ngcf_oracle.py (torch CPU ops of NGCF.py:120-156 + bprloss.py) forward, "
                                         f"torch autograd backward, torch.optim.Adam; no dropout; one run of {dt:.1f} s at {host_cores()} "
                                         f"threads, torch {torch.__version__}; value = nnz(L) / that time"}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def edges_per_step_of(n_layer, nnz):
    return n_layer * nnz


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    # NGCF_BENCH_SHARE_GPU=1 is a rehearsal mode for a one-GPU box: all ranks use cuda:0 over gloo (RCCL refuses
    # two ranks on one device); the driver's runs use one GPU per rank over "nccl" (= RCCL).
    share = os.environ.get("NGCF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    n_vis = torch.cuda.device_count()
    if n_vis and local_rank >= n_vis:      # a launcher that shows every rank its own GPU only (HIP_VISIBLE_DEVICES per rank)
        local_rank %= n_vis
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        import datetime
        os.environ.setdefault("NGCF_P2P_TIMEOUT_MS", "30000")  # a dead peer must not hold the others for the default minute per wait
        to = datetime.timedelta(minutes=5)             # a rank that dies must not leave the others waiting for the default half hour
        if share:
            dist.init_process_group("gloo", timeout=to)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=to)

    import seoul_tourism_recommendation_ngcf_amd as pkg
    from seoul_tourism_recommendation_ngcf_amd import _lib, dist as ngcf_dist
    lib = _lib.load()
    t_start = time.perf_counter()

    note_dir = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"ngcf_bench_{os.environ.get('MASTER_PORT', '0')}")

    def note(msg):          # N > 1: a line per phase on stderr (every rank), so that a slow or failed phase can be found in the logs
        if world > 1:
            line = f"+{time.perf_counter() - t_start:.1f}s {msg}"
            print(f"[bench rank {rank} {line}]", file=sys.stderr, flush=True)
            try:            # ... and in a file per rank: the watchdog's line names every rank's last phase
                os.makedirs(note_dir, exist_ok=True)
                with open(os.path.join(note_dir, f"rank{rank}.note"), "w") as f:
                    f.write(line)
            except OSError:
                pass

    def last_notes():
        out = {}
        for q in range(world):
            try:
                out[str(q)] = open(os.path.join(note_dir, f"rank{q}.note")).read()
            except OSError:
                out[str(q)] = None
        return out
    note("process group up")

    if args.workload == "c1_train":
        if world != 1:
            raise SystemExit("c1_train is a single-GPU configuration")
        return main_train(args, pkg, dev)
    if args.workload == "c3_train":
        if world != 1:
            raise SystemExit("c3_train is a single-GPU configuration")
        return main_c3_train(args, pkg, lib, dev)
    n_user, n_item, n_inter, d0, layers, seed = WORKLOADS[args.workload]
    seoul = args.workload in ("c1", "c2")
    full_forward = d0 % 5 == 0                                # a width the reference accepts: time the whole NGCF.forward
    if full_forward and world != 1:
        raise SystemExit("the full-forward workloads are single-GPU configurations")
    if seoul:
        coo = pkg.graphs.seoul_standin(dev, seed=seed, n_user=n_user, n_item=n_item)[0]
    elif world == 1:
        coo = pkg.graphs.synthetic_bipartite(n_user, n_item, n_inter, seed=seed, device=dev,
                                             item_skew=not args.uniform_items)
    else:
        # every rank draws the same interaction triplets and keeps only its own slabs (dist.from_interactions): the doubled
        # [N, N] COO is never formed on any rank
        inter = pkg.graphs.synthetic_interactions(n_user, n_item, n_inter, seed=seed, device=dev,
                                                  item_skew=not args.uniform_items)
        coo = {"nnz": int(2 * inter[0].numel()), "interactions": int(inter[0].numel()), "n_user": n_user, "n_item": n_item}
    nnz, N = coo["nnz"], n_user + n_item
    num_dict = {"user": n_user, "item": n_item, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    torch.manual_seed(seed)                                   # same parameters on every rank
    lap = pkg.graphs.to_sparse_coo(coo) if world == 1 else None
    model = pkg.NGCF(d0, list(layers), None, None, 1.0, [lap], num_dict, args.batch, dev).to(dev).eval()
    model.check_indices = False                               # ids are generated in range; no host sync per step
    crit = pkg.BPR(0.025, args.batch)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    u_id = torch.randint(0, n_user, (args.batch,), generator=g).to(dev)
    pos = torch.randint(0, n_item, (args.batch,), generator=g).to(dev)
    neg = torch.randint(0, n_item, (args.batch,), generator=g).to(dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    if full_forward:
        csr = model.laplacian_csr(0)
        if args.seg_len:
            csr.plan(args.seg_len)
        local_nnz = csr.nnz
        spmm_shapes = [(csr.nnz, csr.n_rows, csr.n_cols)]
        feats = {k: torch.randint(0, c, (args.batch,), generator=g).to(dev)
                 for k, c in (("age", 76), ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
        year = torch.full((args.batch,), 18, device=dev)

        if args.hipgraph:                             # the batch lives in the graph's static index buffers
            fwd = pkg.GraphedForward(model, args.batch, 0)
            fwd(year=year, u_id=u_id, pos_item=pos, neg_item=neg, node_flag=False, **feats)

        def step():                                   # the whole NGCF.forward incl. the feature injection, then BPR
            if args.hipgraph:
                u, p, n = fwd.replay()
            else:
                u, p, n = model(year=year, u_id=u_id, pos_item=pos, neg_item=neg, node_flag=False, **feats)
            return crit(u, p, n)
    elif world == 1:
        csr = model.laplacian_csr(0)
        if args.seg_len:
            csr.plan(args.seg_len)
        local_nnz = csr.nnz
        spmm_shapes = [(csr.nnz, csr.n_rows, csr.n_cols)]

        def step():
            model.propagate(0)
            u = pkg.engine.gather_rows(model.all_users_emb, u_id, status)
            p = pkg.engine.gather_rows(model.all_items_emb, pos, status)
            n = pkg.engine.gather_rows(model.all_items_emb, neg, status)
            return crit(u, p, n)
    else:
        note("interactions drawn")
        sh = ngcf_dist.ShardedPropagation.from_interactions(model, *inter, mode=args.exchange, device=dev)
        note(f"sharded propagation ready: transport={sh.backend}" + (f" (p2p fell back: {sh.p2p_error})" if getattr(sh, "p2p_error", None) else ""))
        local_nnz = sh.local_nnz
        spmm_shapes = sh.spmm_shapes()

        def step():
            sh.propagate()
            u, p, n = sh.gather(u_id, pos, neg)
            return crit(u, p, n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes as C
    host_issue = [None, None]

    def timed(step_fn, prof):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(args.warmup):
            last = step_fn()
        barrier()
        note("warm-up done")
        if prof:
            lib.ngcf_prof_enable(1)                           # hipEvent pair around every SpMM launch
            if world > 1 and hasattr(sh, "p2p_stats"):
                sh.p2p_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step_fn()
        host_issue[0] = (time.perf_counter() - t0) / args.steps * 1e3      # what the host spent issuing (incl. its waits for peers)
        barrier()
        el = time.perf_counter() - t0
        if prof and world > 1 and hasattr(sh, "p2p_stats"):
            host_issue[1] = sh.p2p_stats()
        n_l, ms = C.c_int64(), C.c_double()
        if prof:
            _lib.check(lib.ngcf_prof_collect(C.byref(n_l), C.byref(ms)))
            lib.ngcf_prof_enable(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(last).item(), "non-finite loss"
        return el, last, n_l, ms

    rccl_ranks = None
    if world > 1 and dist.get_backend() == "nccl":
        try:       # the size of the process group's own RCCL communicator, asked of the library (ngcf_comm_size, include/ngcf_hip.h)
            import ctypes as _C
            comm = dist.group.WORLD._get_backend(dev)._comm_ptr()
            n_r = _C.c_int(0)
            _lib.check(lib.ngcf_comm_size(_C.c_void_p(int(comm)), _C.byref(n_r)))
            rccl_ranks = int(n_r.value)
        except Exception as exc:  # noqa: BLE001
            rccl_ranks = f"unavailable: {exc!r}"[:120]
    torch.set_grad_enabled(False)                             # the metric is the forward pass (inference path)
    dt, loss, n_launch, spmm_ms = timed(step, True)
    headline_host = list(host_issue)
    note(f"headline timed: {dt / args.steps * 1e3:.3f} ms per step")
    n_layer = len(layers)

    def build_line(dt, loss, n_launch, spmm_ms, sh, spmm_shapes, local_nnz):
        """The JSON line of a finished headline measurement (secondaries are attached by the caller)."""
        edges_per_step = n_layer * nnz                            # whole job, all ranks
        value = edges_per_step * args.steps / dt
        # roofline of the dominant kernel (spmm_kernel), this rank: algorithmic bytes per launch / mean duration
        per_launch = sum(spmm_model_a_bytes(z, r_, c_, d0) for z, r_, c_ in spmm_shapes) / len(spmm_shapes)
        mean_ms = spmm_ms.value / max(n_launch.value, 1)
        achieved = per_launch / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")    # PMC-measured HBM bytes per launch (rocprofv3 --pmc)
        if os.path.exists(tpath) and world == 1:
            rec = json.load(open(tpath)).get(args.workload + ("_uniform" if args.uniform_items else ""))
            if rec and rec.get("seg_len", 0) == (args.seg_len or rec.get("seg_len", 0)):
                traffic = rec["hbm_bytes_per_spmm_launch"]
                traffic_source = ("profiles/traffic.json - NOT measured in this run: separate rocprofv3 --pmc passes of this "
                                  "command (tools/pmc.sh), " + rec.get("version", "") + ", " + rec.get("collected", ""))
        gather_bytes = (local_nnz / max(len(spmm_shapes), 1) if world > 1 else nnz) * d0 * 4

        swept = [csr.swept_rows] if world == 1 else sh.swept_rows()
        kernel_name = ("spmm_swept_kernel (one L.E product: a launch per row group + fix-up)" if all(swept) else
                       "spmm_kernel + spmm_sliced_kernel (one L.E product)" if not any(swept) else
                       "spmm_swept_kernel / spmm_kernel (one L.E product, mean over the rank's two products)")
        out = {
            "metric": "NGCF 3-layer forward: propagated edges/sec + achieved HBM GB/s, d=128",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {'Seoul-shaped stand-in' if seoul else 'synthetic bipartite'} {n_user} users x {n_item} items, "
                                   f"{coo['interactions']} interactions, nnz(L)={nnz}, d0={d0}, layers={list(layers)}, "
                                   f"batch={args.batch}, seed={seed}" + (", uniform items" if args.uniform_items else ""),
                       "n_user": n_user, "n_item": n_item, "interactions": coo["interactions"], "nnz_L": nnz,
                       "d": d0, "n_layers": n_layer, "batch": args.batch,
                       "hipgraph": bool(args.hipgraph),
                       "parallelism": "single GPU" if world == 1 else
                       f"row-partition x{world}, exchange={args.exchange} ({ngcf_dist.SCHEME_NOTES[args.exchange]}), transport={sh.backend}"
                       + (f" (p2p fell back: {sh.p2p_error})" if getattr(sh, "p2p_error", None) else ""),
                       **({} if world == 1 else {"exchange": args.exchange, "transport": sh.backend,
                                                 "p2p_mode": getattr(getattr(sh, "p2p", None), "mode", None),
                                                 "rccl_ranks": rccl_ranks, "process_group_backend": dist.get_backend()})},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_launch, "launches_timed": int(n_launch.value),
                         "traffic_source": traffic_source,
                         "mean_launch_ms": mean_ms, "swept_rows": swept},
            # what actually bounds the SpMM on a graph without locality: every stored entry gathers one row slice of E through
            # the vector L1 from L2 (4*d bytes per entry, 26x the algorithmic bytes at C3); see DESIGN.md 4.1
            "roofline_l2": {"bound": "l2-gather", "kernel": kernel_name, "achieved": gather_bytes / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0,
                            "peak": L2_PEAK_GBS, "unit": "GB/s",
                            "frac": (gather_bytes / (mean_ms * 1e-3) / 1e9 / L2_PEAK_GBS) if mean_ms > 0 else 0.0,
                            "gather_bytes_per_launch": gather_bytes,
                            # r04: the same ceiling as a request rate - one 128-byte line request per (entry, line): 16.8-18.8 TB/s
                            # of gathered rows = 131-147 G requests/s (TCC_HIT + TCC_MISS of a launch: profiles/r04_c3_pmc_summary.txt)
                            "line_requests_per_launch": gather_bytes / 128,
                            "achieved_G_lines_per_s": gather_bytes / 128 / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0,
                            "ceiling_G_lines_per_s": [131, 147],
                            "note": "explanatory, not the headline: roofline.frac stays SURVEY 8d model A"},
            "loss": float(loss),
        }
        out["host_issue_ms_per_step"] = headline_host[0]          # rank 0: wall time of the step loop before the closing barrier
        if world > 1:
            st = headline_host[1]                                 # rank 0: host time inside the exchange's waits (CU-free transport only)
            out["p2p_host"] = None if not st else {**st, "host_blocked_ms_per_step": st["host_blocked_ms"] / args.steps,
                                                   "note": "the exchange runs no kernel and moves rows on copy engines; its cost is host "
                                                           "time in bounded waits for the peers' publications - zero when they are there"}
            out["p2p_error"] = getattr(sh, "p2p_error", None) or getattr(sh, "gather_p2p_error", None)   # why the CU-free transport was not used (None: it was, or was not asked for)
            out["secondary_overrun"] = False                      # True only on a line the watchdog printed
        return out

    # N > 1: what follows the headline (the RCCL cross-check, the other exchange scheme) must not be able to lose it: a
    # collective that never completes would otherwise hold every rank until the process group's timeout aborts the job.  A
    # watchdog prints the finished headline and ends the process if the secondaries overrun their budget.
    watchdog = None
    if world > 1 and not args.no_secondary:
        import threading
        budget = float(os.environ.get("NGCF_BENCH_SECONDARY_TIMEOUT_S", "240"))
        fallback = build_line(dt, loss, n_launch, spmm_ms, sh, spmm_shapes, local_nnz)
        fallback["secondary"] = {"error": f"the secondary measurements did not finish within {budget:.0f} s; this line was printed by the watchdog"}
        fallback["cpu_baseline"] = None
        line_lock, line_done = threading.Lock(), [False]

        def overrun():
            with line_lock:
                if line_done[0]:
                    return
                line_done[0] = True
                if rank == 0:
                    fallback["secondary_overrun"] = True
                    fallback["last_phase_per_rank"] = last_notes()
                    print(json.dumps(fallback), flush=True)
                    sys.stdout.flush()
            os._exit(3)            # the line is out; a hang in a secondary is still a failure the harness must see
        watchdog = threading.Timer(budget + (0.0 if rank == 0 else 5.0), overrun)
        watchdog.daemon = True
        watchdog.start()
    transport_check = None
    if world > 1 and sh.backend == "p2p" and not args.no_secondary:
        # The same scheme over torch.distributed collectives (RCCL), timed the same way: a cross-check of the p2p transport's
        # RESULT (the two losses must agree) and its price tag.  `value` is the p2p run unless the losses disagree - then the
        # RCCL run is the headline and the line says so.
        try:       # the headline must not be lost to a failure of the cross-check
            os.environ["NGCF_DIST_COLLECTIVES"] = "torch"
            sh_t = ngcf_dist.ShardedPropagation.from_interactions(model, *inter, mode=args.exchange, device=dev)
            os.environ.pop("NGCF_DIST_COLLECTIVES")

            def step_t():
                sh_t.propagate()
                u, p, n = sh_t.gather(u_id, pos, neg)
                return crit(u, p, n)
            dt_t, loss_t, n_launch_t, spmm_ms_t = timed(step_t, True)
            agree = abs(float(loss) - float(loss_t)) <= 1e-5 * abs(float(loss_t))
            transport_check = {"p2p": {"ms_per_step": dt / args.steps * 1e3, "loss": float(loss)},
                               "torch_rccl": {"ms_per_step": dt_t / args.steps * 1e3, "loss": float(loss_t)}, "losses_agree": bool(agree)}
            if not agree:
                dt, loss, n_launch, spmm_ms, sh = dt_t, loss_t, n_launch_t, spmm_ms_t, sh_t
                spmm_shapes, local_nnz = sh.spmm_shapes(), sh.local_nnz
        except Exception as exc:  # noqa: BLE001
            os.environ.pop("NGCF_DIST_COLLECTIVES", None)
            transport_check = {"p2p": {"ms_per_step": dt / args.steps * 1e3, "loss": float(loss)}, "torch_rccl": {"error": repr(exc)[:300]}}

    secondary = {}
    if world > 1 and not args.no_secondary:
        other = "bipartite" if args.exchange == "allgather" else "allgather"
        try:       # the headline must not depend on the secondary measurement
            sh2 = ngcf_dist.ShardedPropagation.from_interactions(model, *inter, mode=other, device=dev)

            def step2():
                sh2.propagate()
                u, p, n = sh2.gather(u_id, pos, neg)
                return crit(u, p, n)
            dt2, loss2, _, _ = timed(step2, False)
            secondary[f"exchange_{other}"] = {"value": edges_per_step_of(len(layers), nnz) * args.steps / dt2, "unit": "edges/s",
                                              "ms_per_step": dt2 / args.steps * 1e3, "loss": float(loss2), "transport": sh2.backend,
                                              "note": ngcf_dist.SCHEME_NOTES[other]}
            del sh2
        except Exception as exc:  # noqa: BLE001
            secondary[f"exchange_{other}"] = {"error": repr(exc)[:300]}
    if world == 1 and not seoul and not args.no_secondary and d0 % 5 != 0:
        try:       # the headline must not depend on the secondary measurement
            # the widths the reference can actually run (embed_size must be a multiple of 5, NGCF.py:39-43,114): same graph,
            # embed_size = 5*ceil(d0/5) -> [d0]*n, through the whole model.forward() including the feature injection
            d5 = (d0 + 4) // 5 * 5
            torch.manual_seed(seed)
            m5 = pkg.NGCF(d5, list(layers), None, None, 1.0, [lap], num_dict, args.batch, dev).to(dev).eval()
            m5.check_indices = False
            feats = {k: torch.randint(0, c, (args.batch,), generator=g).to(dev)
                     for k, c in (("age", 76), ("sex", 2), ("month", 13), ("day", 32), ("dow", 7))}
            year = torch.full((args.batch,), 18, device=dev)

            def step5():
                u, p, n = m5(year=year, u_id=u_id, pos_item=pos, neg_item=neg, node_flag=False, **feats)
                return crit(u, p, n)
            dt5, loss5, _, _ = timed(step5, False)
            secondary[f"reference_legal_{d5}_to_{d0}"] = {
                "value": len(layers) * nnz * args.steps / dt5, "unit": "edges/s", "ms_per_step": dt5 / args.steps * 1e3,
                "ratio_to_headline": dt5 / dt, "loss": float(loss5),
                "note": f"embed_size={d5}, layer_size={list(layers)}: whole NGCF.forward (feature injection, propagation, "
                        f"gathers) + BPR; the headline runs d0={d0}, which the reference itself cannot (NGCF.py:39-43,114)"}
            del m5
        except Exception as exc:  # noqa: BLE001
            secondary["reference_legal_width"] = {"error": repr(exc)[:300]}

    if world == 1 and not seoul and not args.no_secondary:
        # BASELINE configs 0-1 (Seoul-shaped stand-in graph, 2 layers, d = 64 / 512): the whole NGCF.forward incl. the feature
        # injection + BPR, replayed as a hipGraph - launch-bound sizes, reported beside the headline
        for wl in ("c1", "c2"):
            try:
                nu, ni, _, e0, lay_s, sd_s = WORKLOADS[wl]
                coo_s = pkg.graphs.seoul_standin(dev, seed=sd_s, n_user=nu, n_item=ni)[0]
                nd_s = {"user": nu, "item": ni, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
                torch.manual_seed(sd_s)
                ms = pkg.NGCF(e0, list(lay_s), None, None, 1.0, [pkg.graphs.to_sparse_coo(coo_s)], nd_s, args.batch, dev).to(dev).eval()
                ms.check_indices = False
                gs = torch.Generator(device="cpu").manual_seed(sd_s + 1)
                ids = {k: torch.randint(0, c, (args.batch,), generator=gs).to(dev)
                       for k, c in (("u_id", nu), ("pos_item", ni), ("neg_item", ni), ("age", 76), ("sex", 2), ("month", 13),
                                    ("day", 32), ("dow", 7))}
                crit_s = pkg.BPR(0.025, args.batch)
                fwd_s = pkg.GraphedForward(ms, args.batch, 0, criterion=crit_s)     # forward + BPR: one graph launch per step
                fwd_s(year=torch.full((args.batch,), 18, device=dev), node_flag=False, **ids)

                def step_s():
                    fwd_s.replay()
                    return fwd_s.loss
                med, lo, hi, per, ls = timed_blocks(step_s, blocks=5, steps=40, warmup=20)
                secondary[f"{wl}_seoul_shaped_hipgraph"] = {
                    "ms_per_step": med, "ms_per_step_min": lo, "ms_per_step_max": hi, "blocks_ms_per_step": [round(x, 4) for x in per],
                    "value": len(lay_s) * coo_s["nnz"] / (med * 1e-3), "unit": "edges/s", "loss": float(ls),
                    "note": f"{nu} users x {ni} items, nnz(L)={coo_s['nnz']}, embed_size={e0}, layers={list(lay_s)}, batch={args.batch}: "
                            "NGCF.forward (injection, propagation, gathers) + BPR replayed as ONE hipGraph; five blocks of 40 steps after 20 "
                            "(median / min / max; the garbage collector is off inside a block)"}
                del fwd_s, ms
            except Exception as exc:  # noqa: BLE001
                secondary[f"{wl}_seoul_shaped_hipgraph"] = {"error": repr(exc)[:300]}

    if world == 1 and not seoul and not args.no_secondary:
        # the reference's own training step (main.py:63-76, experiment.py:45-58) on the Seoul-shaped stand-in, in both dropout modes
        torch.set_grad_enabled(True)
        for mode in ("reference", "device"):
            try:
                secondary[f"c1_train_{mode}_masks"] = train_secondary(pkg, dev, args.batch, mode)
            except Exception as exc:  # noqa: BLE001
                secondary[f"c1_train_{mode}_masks"] = {"error": repr(exc)[:300]}
        try:       # the steady state of the reference's own loop: eval() at the end of epoch 1 is never undone (experiment.py:61,72)
            secondary["c1_train_eval_mode_device_masks"] = train_secondary(pkg, dev, args.batch, "device", eval_mode=True)
        except Exception as exc:  # noqa: BLE001
            secondary["c1_train_eval_mode_device_masks"] = {"error": repr(exc)[:300]}
        try:
            secondary["c1_train_device_masks_hipgraph"] = train_secondary(pkg, dev, args.batch, "device", 100, 10, graphed=True)
        except Exception as exc:  # noqa: BLE001
            secondary["c1_train_device_masks_hipgraph"] = {"error": repr(exc)[:300]}
        if args.workload == "c3":
            try:   # the training step at scale on the headline's own graph (own line: --workload c3_train)
                r3 = c3_train_measure(pkg, lib, dev, lap, n_user, n_item, nnz, args.batch, steps=5, warmup=2, seed=seed)
                r3.pop("model"), r3.pop("ids")
                secondary["c3_train_device_masks"] = r3
            except Exception as exc:  # noqa: BLE001
                secondary["c3_train_device_masks"] = {"error": repr(exc)[:300]}
        torch.set_grad_enabled(False)

    out = build_line(dt, loss, n_launch, spmm_ms, sh if world > 1 else None, spmm_shapes, local_nnz)
    if transport_check:
        out["transport_check"] = transport_check
    if secondary:
        out["secondary"] = secondary
    if seoul:
        out["roofline"]["note"] = "working set is cache-resident at this size: the HBM fraction is not meaningful (SURVEY 8d)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and seoul:
        out["cpu_baseline"] = cpu_baseline_full(coo, model, host_cores(), seed)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(coo, model, host_cores(), seed)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and out["cpu_baseline"]:
        out["parity"] = out["cpu_baseline"].pop("parity")     # the oracle as the checker, beside it as the baseline
        assert out["parity"]["ok"], f"engine disagrees with the CPU oracle: {out['parity']}"
    if watchdog is not None:
        with line_lock:
            if line_done[0]:                                  # the watchdog is printing / has printed
                time.sleep(3600)
            line_done[0] = True
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
