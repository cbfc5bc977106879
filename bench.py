#!/usr/bin/env python
"""Headline benchmark: ADMM iterations/s (+ B&B nodes/s) on random_miqp n=500 m=1000 p=250.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the random MIQP of the reference's own generator recipe
(/root/reference/examples/random_miqp/run_example.py:71-83, seed 0, density 0.7) at n=500,
m=1000, 250 binaries, explored by the branch-and-bound host logic one node at a time per GPU with
the reference's settings (run_example.py:98-116).  One "step" = one wave = `--wave` node
relaxations per rank (default 1: node-at-a-time) followed by the incumbent exchange.  When a tree
closes (seed 0 closes after ~220 nodes) the search re-roots on the next MIQP of a stream that shares
P and A -- hence the factor -- and draws new q, l, u, through MIOSQP.update_vectors.  Inputs
(factor, matrices) are resident in HBM before the timed region; the per-node vectors (l, u, x0,
y0: 34 KB) are part of the path and travel inside it.  N > 1 shards the open leaves over the
ranks (miosqp_amd/dist.py), one process per GPU, RCCL only for the incumbent: weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KERNELS = ["k_panel_fwd", "k_tail_fwd", "k_tail_bwd", "k_panel_bwd"]
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/
    r*_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        kernels = json.load(open(files[-1]))["kernels"]
        for name in sorted(kernels):  # template instances are listed as "name<...>"
            if name == kernel or name.startswith(kernel + "<"):
                return kernels[name]["traffic_bytes"]
        return None
    except Exception:
        return None


def large_leg(seed, nodes=12):
    """Extra leg, not part of `value`: BASELINE configs[4] (random_miqp n=5000 m=10000 p=2500, 1 % dense
    A, fp64), where the single-node iteration is HBM-bandwidth-bound (317 MB algorithmic per iteration)."""
    from miosqp_amd import bnb, dist, problems
    cfg = problems.CONFIGS["cfg5"]
    prob = problems.random_miqp(seed=seed, **cfg)
    st = dict(problems.BNB_SETTINGS)
    st["max_iter_bb"] = 10 ** 9
    model = bnb.MIOSQP()
    t0 = time.time()
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], st,
                dict(problems.QP_SETTINGS))
    setup_s = time.time() - t0
    eng = model.work.solver
    srch = dist.ShardedSearch(model)
    srch.step(1)
    eng.loop_stats(reset=True)
    n0, i0 = srch.nodes, srch.iters
    t0 = time.perf_counter()
    for _ in range(nodes):
        srch.step(1)
    dt = time.perf_counter() - t0
    ms, it = eng.loop_stats()
    fs = eng.factor_stats()
    gbs = fs["bytes_per_iter"] * it / max(1e-9, ms) * 1e-6
    return dict(workload="random_miqp n=%d m=%d p=%d density %.2f, node-at-a-time" %
                         (cfg["n"], cfg["m"], cfg["p"], cfg["density"]),
                iters_per_s=round((srch.iters - i0) / dt, 1), nodes_per_s=round((srch.nodes - n0) / dt, 2),
                nnz_L=fs["nnz_L"], setup_s=round(setup_s, 2), bytes_per_iter=fs["bytes_per_iter"],
                usec_per_iter=round(1e3 * ms / max(1, it), 2), achieved_gbs=round(gbs, 1),
                frac=round(gbs / HBM_PEAK_GBS, 4), factor_form="L (4 launches/iteration)")


def cpu_baseline(prob, budget_s):
    """The CPU oracle ("port": own restatement, NOT the real OSQP which is absent) on the same tree,
    one thread, bounded to about `budget_s` seconds."""
    from miosqp_amd import bnb, dist, problems
    from oracle import oracle
    st = dict(problems.BNB_SETTINGS)
    st["max_iter_bb"] = 10 ** 9
    model = bnb.MIOSQP(backend=oracle)
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"],
                prob["i_u"], st, dict(problems.QP_SETTINGS))
    srch = dist.ShardedSearch(model)
    t0 = time.time()
    while time.time() - t0 < budget_s and model.work.leaves:
        srch.step(1)
    dt = time.time() - t0
    return dict(value=srch.iters / dt, unit="ADMM iter/s", cores=1, kind="port",
                nodes_per_s=srch.nodes / dt,
                sample="first %d nodes (%d ADMM iterations, %.1f s) of the same tree, oracle/qp_oracle.c "
                       "single thread; real OSQP is not installed" % (srch.nodes, srch.iters, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--wave", type=int, default=1, help="node relaxations per rank per step")
    ap.add_argument("--step-budget-ms", type=float, default=-1.0,
                    help="with more than one rank a step is 'node relaxations for this long, at least one' "
                         "instead of a fixed count (ranks then meet at the exchange without waiting for the "
                         "rank that drew the expensive node); -1 = 0.5 + 0.5 log2(ranks) ms, 0 = fixed count (--wave)")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg5"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-width", type=int, default=256,
                    help="extra leg: leaves per batched wave (BASELINE configs[2]); 0 = skip")
    ap.add_argument("--batch-waves", type=int, default=12)
    ap.add_argument("--no-large-leg", action="store_true",
                    help="skip the extra BASELINE configs[4] leg (n=5000, bandwidth-bound single-node ADMM)")
    args = ap.parse_args()

    import torch
    from miosqp_amd import bnb, dist, problems

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the relaxation engine has no CPU fallback")
    # MIOSQP_BENCH_ONE_DEVICE=1: every rank uses GPU 0 and the collectives run over gloo (CPU tensors);
    # only for exercising the multi-rank code path on a one-GPU box, never for reported numbers
    one_dev = os.environ.get("MIOSQP_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
        # several processes time-sharing one GPU cannot keep a cooperative launch co-resident (it needs the
        # device to itself, one process per GPU as deployed): this test mode uses the two-kernel form
        os.environ.setdefault("MIOSQP_COOP", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as td
        if one_dev:
            td.init_process_group(backend="gloo")
            comm = dist.TorchComm(torch.device("cpu"))
        else:
            td.init_process_group(backend="nccl", device_id=dev)
            comm = dist.TorchComm(dev)
    else:
        comm = dist.LocalComm()
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    cfg = problems.CONFIGS[args.config]
    prob = problems.random_miqp(seed=args.seed, **cfg)
    st = dict(problems.BNB_SETTINGS)
    st["max_iter_bb"] = 10 ** 9  # fixed node budget comes from --steps, not from the tree
    qs = dict(problems.QP_SETTINGS)
    qs["device"] = local_rank
    qs["max_batch"] = max(64, args.batch_width)
    model = bnb.MIOSQP()
    t_setup = time.time()
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"],
                prob["i_u"], st, qs)
    t_setup = time.time() - t_setup
    eng = model.work.solver
    srch = dist.ShardedSearch(model, comm)
    m_orig = cfg["m"]
    rng = np.random.RandomState(args.seed + 12345)
    stream = dict(instances=1)

    def next_instance():
        """The tree closed: re-root on the next MIQP of the stream.  Same P and A, hence the same
        factor in HBM; new q, l, u drawn like the generator draws them (run_example.py:76-80), pushed
        through MIOSQP.update_vectors exactly like the reference's MPC loop does
        (/root/reference/miosqp/solver.py:174-205).  Every rank draws the same numbers."""
        q = rng.randn(cfg["n"])
        u = 2 + rng.rand(m_orig)
        l = -2 + rng.rand(m_orig)
        model.update_vectors(q=q, l=l, u=u)
        srch.begin_instance()
        stream["instances"] += 1

    def sync():
        comm.barrier()
        torch.cuda.synchronize()

    if args.step_budget_ms < 0:
        import math
        args.step_budget_ms = 0.5 + 0.5 * math.log2(max(1, world))
    budget = 1e-3 * args.step_budget_ms if (world > 1 and args.step_budget_ms > 0) else None

    def run_steps(count, width, batched):
        for _ in range(count):
            if srch.step(width, batched, budget=None if batched else budget) == 0:
                next_instance()

    run_steps(args.warmup, args.wave, False)
    sync()
    eng.loop_stats(reset=True)
    n0, i0, inst0 = srch.nodes, srch.iters, stream["instances"]
    t0 = time.perf_counter()
    run_steps(args.steps, args.wave, False)
    srch.drain()  # the exchange still in flight belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    loop_ms, loop_iters = eng.loop_stats()
    nodes_here = srch.nodes - n0
    tot = comm.sum([srch.iters - i0, srch.nodes - n0, dt])
    dt_max = dt
    if world > 1:
        import torch.distributed as td
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm.device)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt_max = float(tmax.item())
    iters, nodes = float(tot[0]), float(tot[1])
    instances = stream["instances"] - inst0 + 1

    # ---- extra leg (BASELINE configs[2]): the same stream explored in waves of up to `batch_width`
    #      leaves per rank, each wave ONE batched device call (not part of `value`) -------------------
    batched = None
    if args.batch_width > 0:
        next_instance()
        run_steps(12, args.batch_width, True)  # warm-up: graph capture, allocation, frontier ramp-up
        sync()
        eng.batch_stats(reset=True)
        n1, i1 = srch.nodes, srch.iters
        t1 = time.perf_counter()
        run_steps(args.batch_waves, args.batch_width, True)
        srch.drain()
        sync()
        dtb = time.perf_counter() - t1
        bms, bit, bnode = eng.batch_stats()
        totb = comm.sum([srch.iters - i1, srch.nodes - n1])
        if world > 1:
            tb = torch.tensor([dtb], dtype=torch.float64, device=comm.device)
            td.all_reduce(tb, op=td.ReduceOp.MAX)
            dtb = float(tb.item())
        batched = dict(max_wave=args.batch_width, waves=args.batch_waves, nodes=float(totb[1]),
                       mean_wave=round(float(totb[1]) / max(1, args.batch_waves * world), 1),
                       node_iters_per_s=round(float(totb[0]) / dtb, 1), nodes_per_s=round(float(totb[1]) / dtb, 2),
                       lockstep_iters=bit, device_us_per_lockstep_iter=round(1e3 * bms / max(1, bit), 2),
                       device_node_iters_per_s=round(bnode / max(1e-9, bms) * 1e3, 1))

    if rank == 0:
        fs = eng.factor_stats()
        kern = []
        if fs["coop"]:
            # one launch of k_coop IS one node relaxation (all its iterations and tests): average launch
            # from the HIP events around the launches of the timed region
            launches = max(1, nodes_here)
            us, by = 1e3 * loop_ms / launches, fs["bytes_per_iter"] * loop_iters / launches
            kern.append(dict(kernel="k_coop", usec=round(us, 3), bytes=round(by), gbs=round(by / us * 1e-3, 1),
                             launches=launches, iterations_per_launch=round(loop_iters / launches, 1)))
            it_us, it_bytes = eng.time_kernel(4, 2000)
        else:
            names = ["k_fold_fwd", "k_fold_bwd"] if fs["fold"] else KERNELS
            for k, nm in enumerate(names):
                us, by = eng.time_kernel(k, 300)
                kern.append(dict(kernel=nm, usec=round(us, 3), bytes=by, gbs=round(by / us * 1e-3, 1)))
            it_us, it_bytes = eng.time_kernel(4, 100)
        dom = max(kern, key=lambda d: d["usec"])
        roof = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(dom["gbs"] / HBM_PEAK_GBS, 4), traffic=pmc_traffic(dom["kernel"]),
                    traffic_source="profiles/r*_pmc_traffic.json (separate rocprofv3 --pmc passes of this command)",
                    bytes_per_launch=dom["bytes"], usec_per_launch=dom["usec"], kernels=kern,
                    iteration=dict(bytes=fs["bytes_per_iter"],
                                   usec_in_timed_region=round(1e3 * loop_ms / max(1, loop_iters), 3),
                                   usec_back_to_back=round(it_us, 3),
                                   achieved=round(fs["bytes_per_iter"] * loop_iters / max(1e-9, loop_ms) * 1e-6, 1),
                                   frac=round(fs["bytes_per_iter"] * loop_iters / max(1e-9, loop_ms) * 1e-6 /
                                              HBM_PEAK_GBS, 4)),
                    timing="HIP events on the engine's stream")
        out = dict(metric="ADMM iterations/s (random_miqp n=%d m=%d p=%d, node-at-a-time B&B)" %
                          (cfg["n"], cfg["m"], cfg["p"]),
                   value=round(iters / dt_max, 1), unit="ADMM iter/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=round(1e3 * dt_max / args.steps, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   nodes_per_s=round(nodes / dt_max, 2), iters_per_node=round(iters / max(1.0, nodes), 1),
                   config=dict(workload="BASELINE configs[1]: random_miqp n=%d m=%d p=%d density %.2f seed %d, "
                                        "%s per rank per step, leaves sharded over %d GPU(s)" %
                                        (cfg["n"], cfg["m"], cfg["p"], cfg["density"], args.seed,
                                         ("node relaxations for %.1f ms" % args.step_budget_ms) if budget
                                         else "%d node(s)" % args.wave, world),
                               instance=problems.instance_digest(prob), nnz_L=fs["nnz_L"],
                               factor_form="explicit KKT inverse in registers, cooperative launch per node"
                               if fs["coop"] else "product form L^-1 (2 launches/iteration)" if fs["fold"]
                               else "L (4 launches/iteration)",
                               qp_settings=problems.QP_SETTINGS, rho=0.1, setup_s=round(t_setup, 3)),
                   roofline=roof)
        out["config"]["instances_in_timed_region"] = instances
        if batched is not None and batched["lockstep_iters"] > 0:
            bk = []
            bnames = ["kbm_fwd", "kbm_bwd"] if fs["fold"] else ["kb_" + k[2:] for k in KERNELS]
            for k, nm in enumerate(bnames):
                us, by = eng.time_kernel(10 + k, 30)
                bk.append(dict(kernel=nm, usec=round(us, 2), bytes=by, gbs=round(by / us * 1e-3, 1)))
            batched["kernels"] = bk
            out["batched"] = batched
        if world == 1 and not args.no_large_leg and args.config == "cfg2":
            out["config5"] = large_leg(args.seed)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
