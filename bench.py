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
(factor, matrices) are resident in HBM before the timed region.  The loop between two nodes
(choose_leaf, bound_and_branch, prune: /root/reference/miosqp/solver.py:85-123) runs in the C++ host
library on device-resident leaves (miosqp_amd/search.py; a 96-byte record per node crosses PCIe);
--python-loop drives every node from bnb.Workspace in Python instead (per-node vectors l, u, x0, y0:
34 KB each way), and the `python_loop` leg reports that form next to `value`.  N > 1 shards the open leaves over the
ranks (miosqp_amd/dist.py), one process per GPU, RCCL only for the incumbent: weak scaling.

Extra legs in the same JSON line (none of them is part of `value`):
  batched   BASELINE configs[2]: waves of up to 256 leaves per device call
  stream    the same node-at-a-time workload on the STREAMING forms of the engine (factor re-read from memory every
            iteration): the persistent solver (one launch per node) and the two-launch form beside it -- what the
            north star's roofline is about
  config5   BASELINE configs[4]: n=5000, the bandwidth-bound single-node case
  config4   BASELINE configs[3]: the power-converter MPC sequence (40 MIQPs, n=18)
  config1   BASELINE configs[0]: n=50 (the reference's CPU-runnable size), whole trees, with the CPU oracle beside it
  cpu_baseline  the reference's CPU path (real OSQP when importable, else the oracle) on this box's host cores
  python_loop   the headline workload with the reference's control flow unchanged (bnb.Workspace in Python driving
            miosqp_qp_solve_node); `value` itself runs the same loop compiled into the host library

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
ALL_LEGS = ("batched", "stream", "config5", "config4", "config1", "cpu", "pyloop", "rho_auto")


def pmc_traffic(kernel, tag="", nodes_per_launch=None):
    """HBM-side bytes per launch of `kernel` from the newest COMMITTED rocprofv3 PMC summary
    profiles/r*_pmc_traffic<tag>.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the
    gfx950 correction); (bytes, file) or (None, None).  Not measured in this run: counters need the profiler."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic%s.json" % tag)))
    if not files:
        return None, None
    try:
        rec = json.load(open(files[-1]))
        from miosqp_amd import _lib
        if rec.get("source_digest") != _lib.source_digest():
            return None, "profiles/%s is of other device code (source digest %s, now %s): not used" % (
                os.path.basename(files[-1]), rec.get("source_digest"), _lib.source_digest())
        kernels = rec["kernels"]
        for name in sorted(kernels):  # template instances are listed as "name<...>"
            if name == kernel or name.startswith(kernel + "<"):
                if nodes_per_launch is not None and "traffic_bytes_per_node" in kernels[name]:
                    # (a resident launch is as long as its call: the summary's bytes per NODE x this run's nodes per launch)
                    return int(round(kernels[name]["traffic_bytes_per_node"] * nodes_per_launch)), \
                        "profiles/" + os.path.basename(files[-1]) + " (bytes per node x nodes per launch)"
                return kernels[name]["traffic_bytes"], "profiles/" + os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def host_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(cpu_model=model, nproc=os.cpu_count())


def setup_model(prob, qs, st=None, backend=None):
    from miosqp_amd import bnb, problems
    st = dict(problems.BNB_SETTINGS) if st is None else st
    st["max_iter_bb"] = 10 ** 9  # fixed node budget comes from --steps, not from the tree
    model = bnb.MIOSQP(backend=backend)
    t0 = time.time()
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], st, qs)
    return model, time.time() - t0


def _node_run(model, nodes, warm):
    """`nodes` node relaxations of the model's tree, one at a time (HIP events around every ADMM loop): per-iteration
    device time of the timed region, iterations, nodes, wall rates."""
    from miosqp_amd import dist
    eng = model.work.solver
    srch = dist.ShardedSearch(model)
    for _ in range(warm):
        srch.step(1)
    eng.loop_stats(reset=True)
    n0, i0 = srch.nodes, srch.iters
    t0 = time.perf_counter()
    for _ in range(nodes):
        if srch.step(1) == 0:
            break
    dt = time.perf_counter() - t0
    ms, it = eng.loop_stats()
    return dict(usec_per_iter=1e3 * ms / max(1, it), iters=srch.iters - i0, nodes=srch.nodes - n0, dt=dt,
                launches=srch.nodes - n0, loop_ms=ms, loop_iters=it)


def large_leg(seed, device, nodes=12):
    """BASELINE configs[4] (random_miqp n=5000 m=10000 p=2500, 1 % dense A, fp64): the single-node iteration
    is HBM-bandwidth-bound.  The engine's form for this size is the persistent streaming solver on the factor form
    (one launch per node; `four_launch_form`: the same nodes with four launches per iteration beside it).  Two byte
    conventions side by side: SURVEY sec. 8d's (12 B per factor entry, value + index: 317 MB per iteration) and
    what the kernels really request (the dense tail has no index array: 8 B per entry, ~217 MB)."""
    from miosqp_amd import problems
    cfg = problems.CONFIGS["cfg5"]
    prob = problems.random_miqp(seed=seed, **cfg)
    out = None
    for pers in (-1, 1, 0):
        model, setup_s = setup_model(prob, dict(problems.QP_SETTINGS, device=device, pers=pers))
        eng = model.work.solver
        fs = eng.factor_stats()
        r = _node_run(model, nodes, 1)
        us = r["usec_per_iter"]
        rec = dict(iters_per_s=round(r["iters"] / r["dt"], 1), nodes_per_s=round(r["nodes"] / r["dt"], 2),
                   usec_per_iter=round(us, 2), usec_back_to_back=round(eng.time_kernel(4, 100)[0], 2),
                   # `frac`: what the memory system DELIVERS against 8 TB/s -- the bytes the kernels request (8 B per entry of the
                   # dense rows / tiles they read, no index arrays), replaced below by the HBM counters' bytes where a committed
                   # PMC summary of this device code exists; `frac_sec8d`: SURVEY 8d's formula (12 B per entry of L, two sweeps)
                   # over the same time -- algorithmic bytes, above 1 for a form that reads fewer
                   frac=round(fs["bytes_moved_per_iter"] / us * 1e-3 / HBM_PEAK_GBS, 4), frac_source="bytes requested by the kernels",
                   moved_gbs=round(fs["bytes_moved_per_iter"] / us * 1e-3, 1),
                   moved_frac=round(fs["bytes_moved_per_iter"] / us * 1e-3 / HBM_PEAK_GBS, 4),
                   achieved_gbs_sec8d=round(fs["bytes_per_iter"] / us * 1e-3, 1),
                   frac_sec8d=round(fs["bytes_per_iter"] / us * 1e-3 / HBM_PEAK_GBS, 4),
                   setup_s=round(setup_s, 2))
        if fs["pers"]:
            tiles = eng.tail_inverse_tiles()
            kname = "k_pers<false, true>" if tiles else "k_pers<false, false>"  # (the symmetric form is an instantiation of its own)
            rec.update(tail_inverse_tiles=tiles,
                       factor_form=(("sparse panels of L + the dense tail as S^-1 = L22^-T D22^-1 L22^-1, read as a symmetric matrix: "
                                     "%d square tiles on and above the diagonal, one per workgroup, half the bytes of the two triangles "
                                     "(the first rows of every wave's share resident in LDS)" % tiles) if tiles else
                                    "sparse panels of L + the dense tail as S^-1 = L22^-T D22^-1 L22^-1 (the bytes of the two "
                                    "triangles, one dense phase instead of two)" if fs["tail_inverse"] else
                                    "L (sparse panels + the two triangular sweeps of the pre-inverted tail)") +
                                   ", read from memory every iteration, ONE persistent launch per node (%s)" % kname,
                       kernel=kname, launches=r["launches"],
                       iterations_per_launch=round(r["loop_iters"] / max(1, r["launches"]), 1),
                       usec_per_launch=round(1e3 * r["loop_ms"] / max(1, r["launches"]), 1))
            tr, src = pmc_traffic(kname, "_cfg5")
            if tr is not None:
                gbs = tr / (1e3 * r["loop_ms"] / max(1, r["launches"])) * 1e-3
                rec["pmc_traffic"] = dict(bytes_per_launch=tr, source=src, hbm_measured_gbs=round(gbs, 1),
                                          bytes_per_iter=round(tr / max(1.0, r["loop_iters"] / max(1, r["launches"]))))
                rec["frac"] = round(gbs / HBM_PEAK_GBS, 4)
                rec["frac_source"] = "HBM counters (FETCH_SIZE doubled + WRITE_SIZE per launch, %s) / this run's launch time" % src
        else:
            kern = []
            for k, nm in enumerate(KERNELS):
                kus, kby = eng.time_kernel(k, 100)
                kern.append(dict(kernel=nm, usec=round(kus, 2), bytes_sec8d=kby))
            rec.update(factor_form="L (4 launches/iteration)", kernels_back_to_back=kern)
        if out is None:
            out = dict(workload="random_miqp n=%d m=%d p=%d density %.2f, node-at-a-time" %
                                (cfg["n"], cfg["m"], cfg["p"], cfg["density"]),
                       nnz_L=fs["nnz_L"], bytes_per_iter=fs["bytes_per_iter"], bytes_moved_per_iter=fs["bytes_moved_per_iter"],
                       bound="hbm",
                       note="frac: bytes the memory system delivers per second / 8 TB/s (see frac_source); frac_sec8d: SURVEY 8d's "
                            "12 B per factor entry over the same time; moved_frac: the 8 B per entry the dense tail rows / tiles "
                            "request; usec_per_iter: HIP events around the ADMM loops of the timed nodes (tests included), "
                            "usec_back_to_back: iterations only")
            out.update(rec)
            if rec.get("tail_inverse_tiles"):
                out["note"] += ("; frac_sec8d above 1 is not bandwidth above the peak: the kernel reads the symmetric S^-1 = (L22 D22 L22^T)^-1 "
                                "once per pair (i, j) and without indices -- %.0f MB per iteration where SURVEY 8d's formula counts %.0f MB "
                                "for the two sweeps of L"
                                % (fs["bytes_moved_per_iter"] * 1e-6, fs["bytes_per_iter"] * 1e-6))
            if not fs["pers"]:
                eng.close()
                break
        elif fs["pers"]:
            out["triangular_sweeps_form"] = rec
        else:
            out["four_launch_form"] = rec
        eng.close()
    return out


def stream_leg(prob, device, nodes=40):
    """The HBM / L2-STREAMING form of the engine on the headline workload: product-form factor stored once in HBM and
    re-read by every iteration -- what the north star's "achieved GB/s against the 8 TB/s roofline" is about, and the
    form of every problem beyond the cooperative solver's n + M <= 2048.  Headline of the leg: the persistent streaming
    solver (ONE launch per node, k_pers_small at this size); `two_launch_form`: k_fold_fwd / k_fold_bwd, two launches
    per iteration, on the same nodes -- also what a single-launch engine falls back to on a shared device."""
    from miosqp_amd import problems
    out = None
    for pers in (1, 0):
        model, setup_s = setup_model(prob, dict(problems.QP_SETTINGS, device=device, coop=0, pers=pers))
        eng = model.work.solver
        fs = eng.factor_stats()
        r = _node_run(model, nodes, 5)
        us = r["usec_per_iter"]
        rec = dict(iters_per_s=round(r["iters"] / r["dt"], 1), nodes_per_s=round(r["nodes"] / r["dt"], 2),
                   usec_per_iter=round(us, 3), usec_back_to_back=round(eng.time_kernel(4, 1000 if fs["pers"] else 100)[0], 3),
                   achieved_gbs=round(fs["bytes_per_iter"] / us * 1e-3, 1),
                   frac=round(fs["bytes_per_iter"] / us * 1e-3 / HBM_PEAK_GBS, 4), setup_s=round(setup_s, 3))
        if fs["pers"]:
            rec.update(factor_form="product form L^-1 read from memory (L2 at this size) every iteration, ONE persistent "
                                   "launch per node", kernel="k_pers_small", launches=r["launches"],
                       iterations_per_launch=round(r["loop_iters"] / max(1, r["launches"]), 1),
                       usec_per_launch=round(1e3 * r["loop_ms"] / max(1, r["launches"]), 1))
            tr, src = pmc_traffic("k_pers_small", "_persistent")
            if tr is not None:
                rec["pmc_traffic"] = dict(bytes_per_launch=tr, source=src)
        else:
            kern = []
            for k, nm in enumerate(["k_fold_fwd", "k_fold_bwd"]):
                kus, kby = eng.time_kernel(k, 300)
                krec = dict(kernel=nm, usec=round(kus, 3), bytes=kby, gbs=round(kby / kus * 1e-3, 1))
                tr, src = pmc_traffic(nm, "_two_kernel_form")
                if tr is not None:
                    krec.update(traffic=tr, hbm_measured_gbs=round(tr / kus * 1e-3, 1), traffic_source=src)
                kern.append(krec)
            rec.update(factor_form="product form L^-1 in HBM, 2 launches/iteration", kernels_back_to_back=kern)
        if out is None:
            out = dict(workload="the headline workload on the streaming engine forms (coop=0)", bound="hbm",
                       bytes_per_iter=fs["bytes_per_iter"],
                       note="usec_per_iter: HIP events around the ADMM loops of the timed nodes (tests included); "
                            "usec_back_to_back: iterations only")
            out.update(rec)
            if not fs["pers"]:
                eng.close()
                break
        else:
            out["two_launch_form"] = rec
        eng.close()
    return out


def mpc_leg(device, repeats=3):
    """BASELINE configs[3]: the power-converter MPC sequence (horizon N=3, n=18, 45 rows): 40 consecutive MIQPs
    sharing one factorisation through update_vectors + set_x0, replayed from the committed fixture exactly as
    the reference's closed loop drives MIOSQP (power_converter.py:467-476).  Many tiny sequential nodes
    (~50 iterations each): latency, not bandwidth -- reported as time per node / per MPC step."""
    from miosqp_amd import problems, qp
    pc = problems.load_power_converter()
    pc["qp_settings"] = dict(pc["qp_settings"], device=device)
    recs, model = problems.run_power_converter(pc, qp)  # warm-up + parity check
    ok = all(r["nodes"] == int(pc["nodes"][k]) and r["osqp_iter"] == int(pc["osqp_iter"][k])
             for k, r in enumerate(recs))
    eng = model.work.solver
    eng.loop_stats(reset=True)
    nodes = iters = 0
    t0 = time.perf_counter()
    for _ in range(repeats):
        recs, model = problems.run_power_converter(pc, qp, model=model)
        nodes += sum(r["nodes"] for r in recs)
        iters += sum(r["osqp_iter"] for r in recs)
    dt = time.perf_counter() - t0
    ms, it = eng.loop_stats()
    fs = eng.factor_stats()
    steps = repeats * len(recs)
    out = dict(workload="power_converter MPC N=3 (n=18, 45 rows), %d MPC steps replayed %d times" % (len(recs), repeats),
               engine_form=("one wavefront per tree, explicit KKT inverse in registers (k_tree_w)"
                            if fs["resident"] and model.work.data.n + model.work.data.m + model.work.data.n_int <= 64
                            else "LDS-resident single workgroup") if fs["resident"] else
                           "cooperative" if fs["coop"] else "multi-kernel",
               mpc_steps_per_s=round(steps / dt, 1), nodes_per_s=round(nodes / dt, 1),
               iters_per_s=round(iters / dt, 1), usec_per_node=round(1e6 * dt / max(1, nodes), 1),
               usec_per_mpc_step=round(1e6 * dt / steps, 1), nodes_per_mpc_step=round(nodes / steps, 2),
               iters_per_node=round(iters / max(1, nodes), 1),
               device_usec_per_iter=round(1e3 * ms / max(1, it), 3),
               device_usec_per_node=round(1e3 * ms / max(1, nodes), 1),
               matches_reference_fixture=bool(ok),
               bound="latency (8 KB of factor: LDS-resident, the HBM roofline does not apply)")
    # the same 40 MIQPs as ONE launch of 40 single-wavefront trees (MIOSQP.solve_many / miosqp_qp_solve_trees): what
    # independent instances on one factorisation cost when the chip is not left to one wavefront -- e.g. a batch of
    # scenarios per sampling instant; the closed loop itself is sequential (each step needs the previous one's input)
    inst = [dict(q=pc["q"][k].copy(), l=pc["l"].copy(), u=pc["u"][k].copy(), x0=pc["x0"][k].copy()) for k in range(len(pc["q"]))]
    got = model.solve_many(inst)  # warm-up + parity
    okb = all(g["nodes"] == int(pc["nodes"][k]) and g["osqp_iter"] == int(pc["osqp_iter"][k]) for k, g in enumerate(got))
    reps_b = 20
    tb = time.perf_counter()
    for _ in range(reps_b):
        inst_b = inst * 4  # 160 trees per launch
        got = model.solve_many(inst_b)
    dtb = time.perf_counter() - tb
    nb = reps_b * len(inst_b)
    out["batched"] = dict(what="MIOSQP.solve_many: %d independent MIQPs per launch, one wavefront each" % len(inst_b),
                          mpc_steps_per_s=round(nb / dtb, 1), usec_per_mpc_step=round(1e6 * dtb / nb, 2),
                          nodes_per_s=round(sum(g["nodes"] for g in got) * reps_b / dtb, 1),
                          matches_reference_fixture=bool(okb))
    eng.close()
    return out


def small_leg(seed, device, instances=40):
    """BASELINE configs[0] (n=50, m=100, p=10: the reference's own CPU-runnable size): a sequence of MIQPs on one
    factor (update_vectors like the reference's MPC loop), each tree closed by MIOSQP.solve() -- on this engine one
    launch per tree (k_tree; its ADMM loop on the explicit KKT inverse in the workgroup's registers) -- and, beside
    it, the same sequence on the CPU oracle (test infrastructure used as the CPU yardstick, one thread)."""
    from miosqp_amd import bnb, problems
    cfg = problems.CONFIGS["cfg1"]
    prob = problems.random_miqp(**cfg, seed=seed)
    res = {}
    for name in ("hip", "cpu_oracle"):
        backend = None
        if name == "cpu_oracle":
            from oracle import oracle
            backend = oracle
        model = bnb.MIOSQP(backend=backend) if backend is not None else bnb.MIOSQP()
        qs = dict(problems.QP_SETTINGS)
        if backend is None:
            qs["device"] = device
        model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"],
                    dict(problems.BNB_SETTINGS), qs)
        rng = np.random.RandomState(seed + 4242)
        nodes = iters = 0
        uppers = []
        t0 = None
        for k in range(instances + 3):
            if k == 3:
                t0 = time.perf_counter()  # three warm-up instances
                nodes = iters = 0
            model.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(cfg["m"]), u=2 + rng.rand(cfg["m"]))
            r = model.solve()
            nodes += model.work.iter_num - 1
            iters += model.work.osqp_iter
            if k >= 3:
                uppers.append(float(r.upper_glob))
        dt = time.perf_counter() - t0
        res[name] = dict(miqps_per_s=round(instances / dt, 1), usec_per_miqp=round(1e6 * dt / instances, 1),
                         nodes_per_s=round(nodes / dt, 1), iters_per_s=round(iters / dt, 1),
                         nodes_per_miqp=round(nodes / instances, 2), iters_per_node=round(iters / max(1, nodes), 1))
        res[name + "_uppers"] = uppers
        if backend is None:
            model.work.solver.close()
    # the same sequence as batches of 64 trees per launch (MIOSQP.solve_many: a workgroup per MIQP)
    model = bnb.MIOSQP()
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS, device=device))
    rng = np.random.RandomState(seed + 4242)
    inst = [dict(q=rng.randn(cfg["n"]), l=-2 + rng.rand(cfg["m"]), u=2 + rng.rand(cfg["m"])) for _ in range(instances + 3)]
    Bt = 64
    many = (inst[3:] * ((4 * Bt) // max(1, instances) + 1))[:4 * Bt]
    model.solve_many(many[:Bt])  # warm-up
    t0 = time.perf_counter()
    got = []
    for k in range(0, len(many), Bt):
        got += model.solve_many(many[k:k + Bt])
    dtm = time.perf_counter() - t0
    same_b = all(abs(g["upper_glob"] - y) <= 1e-6 * max(1.0, abs(y)) or (not np.isfinite(g["upper_glob"]) and not np.isfinite(y))
                 for g, y in zip(got[:instances], res["hip_uppers"]))
    res["hip_batched"] = dict(what="MIOSQP.solve_many, %d MIQPs per launch (a workgroup each)" % Bt,
                              miqps_per_s=round(len(many) / dtm, 1), usec_per_miqp=round(1e6 * dtm / len(many), 1),
                              nodes_per_s=round(sum(g["nodes"] for g in got) / dtm, 1), same_optima_as_one_at_a_time=bool(same_b))
    model.work.solver.close()
    a, b = res.pop("hip_uppers"), res.pop("cpu_oracle_uppers")
    same = all((not np.isfinite(x) and not np.isfinite(y)) or abs(x - y) <= 1e-6 * max(1.0, abs(y)) for x, y in zip(a, b))
    return dict(workload="random_miqp n=50 m=100 p=10 density 0.7, %d MIQPs on one factor (update_vectors), whole trees"
                         % instances, hip=res["hip"], hip_batched=res["hip_batched"],
                batched_speedup_over_one_host_core=round(res["hip_batched"]["miqps_per_s"] / max(1e-9, res["cpu_oracle"]["miqps_per_s"]), 2),
                cpu_baseline=dict(res["cpu_oracle"], kind="port", cores=1,
                                  note="oracle/qp_oracle.c (own CPU restatement, not OSQP), the same sequence"),
                same_optima=bool(same),
                speedup_over_one_host_core=round(res["hip"]["miqps_per_s"] / max(1e-9, res["cpu_oracle"]["miqps_per_s"]), 2))


def cpu_baseline(prob, budget_s):
    """The reference's CPU path on this box's host cores, bounded to about `budget_s` seconds of the same tree.
    Probes `import osqp` first: if the real package is present the tree search runs on it
    (kind "reference", with the frozen spec's parameters so that both sides do the same work); otherwise on
    the CPU oracle (kind "port": own restatement of the algorithm, NOT OSQP).  One thread either way: OSQP's
    QDLDL path and the oracle are sequential codes."""
    from miosqp_amd import bnb, dist, problems
    info = host_info()
    kind, backend, why = "port", None, ""
    try:
        import osqp  # noqa: F401  (/root/reference/miosqp/node.py:2)
        if hasattr(osqp, "constant") and hasattr(osqp, "OSQP"):
            backend, kind = osqp, "reference"
        else:
            why = "an `osqp` module is importable but lacks the 0.6-era surface (constant/OSQP) the reference uses"
    except Exception as e:  # absent on the GPU box: it receives only this repository
        why = "import osqp failed (%s)" % type(e).__name__
    qs = dict(problems.QP_SETTINGS)
    if backend is None:
        from oracle import oracle
        backend = oracle
    else:
        qs.update(rho=0.1, sigma=1e-6, alpha=1.6, adaptive_rho=False, max_iter=4000, scaling=10,
                  check_termination=25, polish=False, eps_dual_inf=1e-4)
    try:
        model, _ = setup_model(prob, qs, backend=backend)
    except Exception as e:
        if kind == "reference":  # e.g. an osqp >= 1.0 whose keyword names differ: fall back, and say so
            from oracle import oracle
            why = "real osqp present but its setup() refused the reference's call (%s)" % type(e).__name__
            kind, backend, qs = "port", oracle, dict(problems.QP_SETTINGS)
            model, _ = setup_model(prob, qs, backend=backend)
        else:
            raise
    srch = dist.ShardedSearch(model)
    t0 = time.time()
    while time.time() - t0 < budget_s and model.work.leaves:
        srch.step(1)
    dt = time.time() - t0
    what = "real OSQP %s" % getattr(backend, "__version__", "?") if kind == "reference" else \
        "oracle/qp_oracle.c (own CPU restatement, not OSQP; %s)" % why
    return dict(value=srch.iters / dt, unit="ADMM iter/s", cores=1, kind=kind, nodes_per_s=srch.nodes / dt,
                cpu_model=info["cpu_model"], nproc=info["nproc"],
                sample="first %d nodes (%d ADMM iterations, %.1f s) of the same tree, one thread of %d host cores; %s"
                       % (srch.nodes, srch.iters, dt, info["nproc"] or 0, what))


def rho_auto_leg(prob, cfg, seed, local_rank, nodes, batched_chunks=0):
    """The headline workload with rho chosen ONCE per MIQP at setup (qp setting rho="auto": OSQP's own update rule applied
    to the root's iterates, then frozen -- one factor for every node as before) next to the frozen default rho = 0.1 of
    `value`.  The reference passes only eps_* to osqp.setup (/root/reference/miosqp/workspace.py:67-68), i.e. OSQP's
    defaults, which adapt rho.  Same MIQP stream as the headline (same seed); hosted node-at-a-time search."""
    import torch
    from miosqp_amd import bnb, problems, search
    t0 = time.perf_counter()
    model = bnb.MIOSQP()
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
    model.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], st,
                dict(problems.QP_SETTINGS, device=local_rank, rho="auto"))
    t_setup = time.perf_counter() - t0
    eng = model.work.solver
    hs = search.HostedSearch(model)
    rng = np.random.RandomState(seed + 12345)
    m_orig = cfg["m"]

    def reroot():
        model.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(m_orig), u=2 + rng.rand(m_orig))
        hs.begin_instance()

    # one whole tree (the seed's own MIQP), timed from its root
    torch.cuda.synchronize()
    tt = time.perf_counter()
    n0, i0 = hs.nodes, hs.iters
    alive = 1
    while alive != 0 and time.perf_counter() - tt < 20.0:
        alive = hs.step(256)
    torch.cuda.synchronize()
    tree = dict(ms_to_close=round(1e3 * (time.perf_counter() - tt), 3), nodes=hs.nodes - n0, iters=hs.iters - i0,
                closed=bool(alive == 0), upper_glob=float(model.work.upper_glob))
    # ... and the SAME MIQP under the frozen default rho = 0.1 (an engine of its own, set up and closed here): the pair
    # `by_rho` quotes for the time to close a tree -- like for like, the optima checked against each other
    m01 = bnb.MIOSQP()
    m01.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], dict(st),
              dict(problems.QP_SETTINGS, device=local_rank))
    h01 = search.HostedSearch(m01)
    torch.cuda.synchronize()
    tt = time.perf_counter()
    alive = 1
    while alive != 0 and time.perf_counter() - tt < 20.0:
        alive = h01.step(256)
    torch.cuda.synchronize()
    tree01 = dict(ms_to_close=round(1e3 * (time.perf_counter() - tt), 3), nodes=h01.nodes, iters=h01.iters,
                  closed=bool(alive == 0), upper_glob=float(m01.work.upper_glob))
    m01.work.solver.close()
    same = dict(what="the seed's own MIQP closed from its root under both settings of rho (same instance, same search rule)",
                rho_default=tree01, rho_auto=tree,
                upper_glob_agree=bool(abs(tree01["upper_glob"] - tree["upper_glob"]) <= 1e-3 * max(1.0, abs(tree["upper_glob"]))))
    # the rate over `nodes` node relaxations of the MIQP stream
    reroot()
    left = 20
    while left > 0:
        b = hs.nodes
        if hs.step(left) == 0:
            reroot()
        left -= max(1, hs.nodes - b)
    torch.cuda.synchronize()
    n0, i0 = hs.nodes, hs.iters
    t1 = time.perf_counter()
    left = nodes
    while left > 0:
        b = hs.nodes
        if hs.step(left) == 0:
            reroot()
        left -= max(1, hs.nodes - b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    dn, di = hs.nodes - n0, hs.iters - i0
    out = dict(what="the headline's search with rho chosen once at setup (rho=\"auto\"), then frozen; opt-in",
               rho=eng.rho(), rho_default=0.1, setup_s=round(t_setup, 3), nodes=dn, value=round(di / dt, 1),
               unit="ADMM iter/s", nodes_per_s=round(dn / dt, 2), iters_per_node=round(di / max(1, dn), 1),
               ms_per_node=round(1e3 * dt / max(1, dn), 4), one_tree=tree, same_miqp=same)
    if batched_chunks > 0:
        # configs[2] with the same rho: the stream on the device-resident leaf pool (the `batched` leg's form), 256 columns
        from miosqp_amd import stream as stream_mod
        model.work.leaves = []
        model.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(m_orig), u=2 + rng.rand(m_orig))
        ss = stream_mod.NativeStreamSearch(model, columns=256)
        closed = []

        def chunks(count):
            for _ in range(count):
                if ss.step() == 0:
                    closed.append((time.perf_counter(), ss.nodes))
                    model.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(m_orig), u=2 + rng.rand(m_orig))
                    ss.begin_instance()

        chunks(max(40, batched_chunks // 5))
        torch.cuda.synchronize()
        eng.batch_stats(reset=True)
        del closed[:]
        n2, i2, t2 = ss.nodes, ss.iters, time.perf_counter()
        chunks(batched_chunks)
        ss.step(rounds=-1)  # (the launches in flight: waited for, their digests absorbed and counted)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t2
        sms, sit, snode = eng.batch_stats()
        bt = dict(width=256, chunks=batched_chunks, nodes=ss.nodes - n2, node_iters_per_s=round((ss.iters - i2) / dts, 1),
                  nodes_per_s=round((ss.nodes - n2) / dts, 2), iters_per_node=round((ss.iters - i2) / max(1, ss.nodes - n2), 1),
                  device_us_per_lockstep_iter=round(1e3 * sms / max(1, sit), 2),
                  column_occupancy=round(snode / float(max(1, 256 * sit)), 3), closed_in_timed_region=len(closed))
        if len(closed) >= 2:
            gaps = [closed[k][0] - closed[k - 1][0] for k in range(1, len(closed))]
            bt.update(mean_ms_to_close=round(1e3 * float(np.mean(gaps)), 3),
                      mean_nodes_per_tree=round(float(np.mean([closed[k][1] - closed[k - 1][1] for k in range(1, len(closed))])), 1))
        out["batched"] = bt
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--wave", type=int, default=1, help="node relaxations per rank per step")
    ap.add_argument("--step-budget-ms", type=float, default=-1.0,
                    help="with more than one rank a step is 'node relaxations for this long, at least one' "
                         "instead of a fixed count (ranks then meet at the exchange without waiting for the "
                         "rank that drew the expensive node); -1 = 1.5 + 0.5 log2(ranks per tree) ms, 0 = fixed count (--wave)")
    ap.add_argument("--ranks-per-tree", type=int, default=-1,
                    help="headline with more than one rank: how many ranks share the leaves of ONE tree (incumbent all-gather "
                         "and leaf hand-over inside that group: a sub-communicator); the groups work on different MIQPs of the "
                         "stream at the same time.  -1 = 2 when the number of ranks is even, else all of them; 0 = all ranks "
                         "on one tree.  A tree of config 2 has ~220 nodes: its ramp-up and tail keep 0.87 of two ranks busy "
                         "and 0.68 of eight (profiles/r06_sim_sharded_hosted.txt)")
    ap.add_argument("--python-loop", action="store_true",
                    help="headline: drive every node from Python (solve_node + bnb.Workspace, vectors over PCIe) instead "
                         "of the C++ host loop on device-resident leaves (miosqp_qp_search_*)")
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg5", "cfg5x"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-width", type=int, default=256,
                    help="extra leg: leaves per batched wave (BASELINE configs[2]); 0 = skip")
    ap.add_argument("--batch-waves", type=int, default=6, help="waves of the wave form in the batched leg (0: the stream alone)")
    ap.add_argument("--stream-warmup", type=int, default=700, help="streaming leg: chunks before the timed stretch")
    ap.add_argument("--stream-chunks", type=int, default=600, help="streaming leg: timed chunks")
    ap.add_argument("--pools", type=int, default=2,
                    help="streaming leg, one rank: also run the tree on this many pools of the one GPU (1 = skip)")
    ap.add_argument("--stream-driver", choices=["native", "python"], default="native",
                    help="host side of the streaming search: compiled into the library (miosqp_qp_stream_*) or stream.StreamSearch")
    ap.add_argument("--stream-exchange", type=int, default=4,
                    help="streaming leg with more than one rank: chunks between two exchanges")
    ap.add_argument("--no-large-leg", action="store_true",
                    help="skip the extra BASELINE configs[4] leg (n=5000, bandwidth-bound single-node ADMM)")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the back-to-back kernel timing launches (profiling runs: every dispatch of the hot "
                         "kernel is then a node relaxation of the timed workload)")
    ap.add_argument("--legs", default="all",
                    help="comma list of extra legs to run: batched,stream,config5,config4,config1,cpu,pyloop,rho_auto ('none' = headline only)")
    args = ap.parse_args()
    legs = set(ALL_LEGS) if args.legs == "all" else set(x for x in args.legs.split(",") if x and x != "none")
    if args.no_cpu_baseline:
        legs.discard("cpu")
    if args.no_large_leg:
        legs.discard("config5")
    if args.batch_width <= 0:
        legs.discard("batched")

    import torch
    from miosqp_amd import dist, problems

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run
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
    td = None
    if world > 1 or launched:
        # also with ONE rank when launched through torch.distributed.run: the process group is RCCL and the
        # barrier / all-reduce / all-gather below run on device tensors exactly as they do with 8 ranks
        import torch.distributed as td
        if one_dev:
            td.init_process_group(backend="gloo")
            comm = dist.TorchComm(torch.device("cpu"))
        else:
            td.init_process_group(backend="nccl", device_id=dev)
            comm = dist.TorchComm(dev)
    else:
        comm = dist.LocalComm()
    # ranks per tree: the leaf-sharded search of ONE tree runs inside a group of ranks (its own communicator); the groups take
    # different MIQPs of the stream.  Every rank creates every group (new_group is collective), in the same order.
    rpt = args.ranks_per_tree
    if rpt < 0:
        rpt = 2 if (world >= 2 and world % 2 == 0) else world
    if rpt == 0 or rpt > world or world % max(1, rpt) != 0:
        rpt = world
    tree_group, tree_comm = rank // max(1, rpt), comm
    if td is not None and world > 1 and rpt < world:
        for g in range(world // rpt):
            members = list(range(g * rpt, (g + 1) * rpt))
            grp = td.new_group(ranks=members, backend="gloo" if one_dev else "nccl")
            if g == tree_group:
                tree_comm = dist.TorchComm(torch.device("cpu") if one_dev else dev, group=grp, ranks=members)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    cfg = problems.CONFIGS[args.config]
    prob = problems.random_miqp(seed=args.seed, **cfg)
    qs = dict(problems.QP_SETTINGS)
    qs["device"] = local_rank
    qs["max_batch"] = max(64, args.batch_width)
    model, t_setup = setup_model(prob, qs)
    eng = model.work.solver
    srch = dist.ShardedSearch(model, comm)
    m_orig = cfg["m"]
    # (every rank of a tree group draws the same numbers; the groups draw different ones: different MIQPs of the stream)
    rng = np.random.RandomState(args.seed + 12345 + 7919 * (tree_group if rpt < world else 0))
    stream = dict(instances=1, closed=[])  # closed: (time, global nodes of that tree) per closed tree

    def next_instance(who=None):
        """The tree closed: re-root on the next MIQP of the stream.  Same P and A, hence the same
        factor in HBM; new q, l, u drawn like the generator draws them (run_example.py:76-80), pushed
        through MIOSQP.update_vectors exactly like the reference's MPC loop does
        (/root/reference/miosqp/solver.py:174-205).  Every rank draws the same numbers."""
        who = srch if who is None else who
        stream["closed"].append((time.perf_counter(), who.global_nodes))
        q = rng.randn(cfg["n"])
        u = 2 + rng.rand(m_orig)
        l = -2 + rng.rand(m_orig)
        model.update_vectors(q=q, l=l, u=u)
        who.begin_instance()
        stream["instances"] += 1

    def sync():
        comm.barrier()
        torch.cuda.synchronize()

    if args.step_budget_ms < 0:
        import math
        args.step_budget_ms = 1.5 + 0.5 * math.log2(max(1, rpt))
    budget = 1e-3 * args.step_budget_ms if (world > 1 and args.step_budget_ms > 0) else None

    def run_steps(count, width, batched):
        for _ in range(count):
            if srch.step(width, batched, budget=None if batched else budget) == 0:
                next_instance()

    # the headline: node-at-a-time branch and bound.  The loop runs in the C++ host library on device-resident leaves
    # (miosqp_amd/search.py); with more than one rank every rank runs it on its share of the tree and the ranks meet
    # after every step for the incumbent (dist.ShardedStream).  --python-loop: the same search driven from Python.
    hosted = hasattr(eng, "search_create") and not args.python_loop

    class Head(object):
        """what the timed loop needs of either form"""
        def __init__(self):
            from miosqp_amd import search
            self.hs = search.HostedSearch(model)
            self.sh = None
            if world > 1 or (launched and os.environ.get("MIOSQP_FORCE_EXCHANGE") == "1"):
                # one leaf per rank ends the replicated start-up (a node-at-a-time rank needs one; dry ranks are fed
                # at the exchanges)
                self.sh = dist.ShardedStream(model, tree_comm, search=self.hs, exchange_every=1, ramp_leaves=1,
                                             step_kwargs=dict(nodes=10 ** 9 if budget else args.wave, budget=budget))
            self._g0 = 0

        nodes = property(lambda self: self.hs.nodes)
        iters = property(lambda self: self.hs.iters)
        global_nodes = property(lambda self: self.sh.global_nodes if self.sh else self.hs.nodes - self._g0)

        steps_done = 0
        idle_steps = 0  # steps in which this rank solved no node (it had no leaf: ran dry between two exchanges)

        def step(self):
            before = self.hs.nodes
            alive = self.sh.step() if self.sh else self.hs.step(args.wave)
            self.steps_done += 1
            self.idle_steps += 1 if self.hs.nodes == before else 0
            return alive

        def begin_instance(self):
            self._g0 = self.hs.nodes
            return self.sh.begin_instance() if self.sh else self.hs.begin_instance()

        def drain(self):
            pass

    if hosted:
        head = Head()
        if rpt < world and tree_group > 0:
            next_instance(head)  # (the groups start on different MIQPs of the stream; group 0 on the seed's own)

        def head_steps(count):
            if head.sh is None:
                # one rank: the `count` steps (one node relaxation each, --wave nodes with that flag) are asked of the
                # library in as few calls as the trees allow -- the interpreter is not between two nodes
                left = count * args.wave
                while left > 0:
                    before = head.hs.nodes
                    alive = head.hs.step(left)
                    left -= head.hs.nodes - before
                    if alive == 0:
                        next_instance(head)
                return
            for _ in range(count):
                if head.step() == 0:
                    next_instance(head)
    else:
        head = srch

        def head_steps(count):
            run_steps(count, args.wave, False)

    head_steps(args.warmup)
    sync()
    # (the warm-up's launches of the solver kernel: with the timed ones they are all a `--legs none --no-probes` process
    #  launches -- what the rocprofv3 kernel table of that command adds up)
    wu_ms, wu_iters = eng.loop_stats() if hasattr(eng, "loop_stats") else (0.0, 0)
    wu_launches = eng.loop_launches() if hasattr(eng, "loop_launches") else 0
    wu_nodes = head.nodes
    eng.loop_stats(reset=True)
    if hosted:
        head._s0, head._d0 = head.steps_done, head.idle_steps
    n0, i0, inst0 = head.nodes, head.iters, stream["instances"]
    del stream["closed"][:]
    t0 = time.perf_counter()
    head_steps(args.steps)
    head.drain()  # the exchange still in flight belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    loop_ms, loop_iters = eng.loop_stats()
    loop_launches = eng.loop_launches() if hasattr(eng, "loop_launches") else 0
    # (which kernel those launches were: the engine says so -- with one node per call a resident launch is also one per node)
    grid_resident = bool(eng.factor_stats().get("search_grid_resident", False))
    node_us = eng.node_stats() if hasattr(eng, "node_stats") else None  # (min, median, max us per iteration over the nodes, count)
    nodes_here = head.nodes - n0
    tot = comm.sum([head.iters - i0, head.nodes - n0, dt])
    # per rank: what it did in the timed region (one row per rank, by all-gather: the collective the search itself uses)
    per_rank = None
    if hosted and world > 1:
        s0, d0 = getattr(head, "_s0", 0), getattr(head, "_d0", 0)
        per_rank = comm.gather([float(rank), float(head.nodes - n0), float(head.iters - i0), float(head.steps_done - s0),
                                float(head.idle_steps - d0), float(getattr(head.sh, "moved", 0)), dt, float(tree_group)])
    if hosted:
        model.work.leaves = []  # the open leaves of this instance live in device slots
    if rpt < world:
        rng.seed(args.seed + 54321)  # (the legs below run on the WHOLE job's communicator: every rank draws the same MIQPs again)
    dt_max = dt
    if td is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm.device)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt_max = float(tmax.item())
    iters, nodes = float(tot[0]), float(tot[1])
    instances = stream["instances"] - inst0 + 1
    # time to close a tree: what branch and bound buys.  Speculative nodes of a parallel search count in `value`
    # (work per second) but only shorten this if they were useful.
    closed = list(stream["closed"])
    trees = dict(closed_in_timed_region=len(closed))
    if len(closed) >= 2:
        gaps = [closed[k][0] - closed[k - 1][0] for k in range(1, len(closed))]
        trees.update(mean_ms_to_close=round(1e3 * float(np.mean(gaps)), 3),
                     mean_nodes_per_tree=round(float(np.mean([c[1] for c in closed[1:]])), 1),
                     trees_per_s=round(1.0 / float(np.mean(gaps)), 2))

    # A driver-sized run (--steps 20 .. 150) ends before the first tree of the timed region closes (seed 0: ~220 nodes):
    # one more tree, from its root to the end, timed on its own -- what branch and bound buys, next to the rates
    # (config 2 only: a tree of the n = 5000 / 8000 shapes is thousands of 15-40 ms nodes; never more than 20 s; not with
    #  --no-probes: the profiled node-only runs hold the launches of the timed workload and nothing else)
    if hosted and world == 1 and len(closed) < 2 and args.config == "cfg2" and not args.no_probes:
        next_instance(head)
        torch.cuda.synchronize()
        tn0, ti0 = head.hs.nodes, head.hs.iters
        tt0 = time.perf_counter()
        open_left = 1
        while open_left != 0 and time.perf_counter() - tt0 < 20.0:
            open_left = head.hs.step(256)
        torch.cuda.synchronize()
        tdt = time.perf_counter() - tt0
        trees["one_tree_after_the_timed_region"] = dict(ms_to_close=round(1e3 * tdt, 3), nodes=head.hs.nodes - tn0,
                                                       iters=head.hs.iters - ti0, closed=bool(open_left == 0),
                                                       upper_glob=float(model.work.upper_glob))
        next_instance(head)

    # ---- extra leg: the same node-at-a-time workload with the reference's own control flow -- bnb.Workspace in
    #      Python driving miosqp_qp_solve_node, vectors over PCIe per node -- next to the hosted loop of `value`
    pyloop = None
    if hosted and "pyloop" in legs and world == 1:
        next_instance()
        run_steps(10, args.wave, False)
        sync()
        np0, ip0 = srch.nodes, srch.iters
        tp = time.perf_counter()
        run_steps(min(args.steps, 100), args.wave, False)
        srch.drain()
        sync()
        dtp = time.perf_counter() - tp
        pyloop = dict(what="bnb.Workspace (Python) driving miosqp_qp_solve_node: the reference's control flow unchanged",
                      value=round((srch.iters - ip0) / dtp, 1), unit="ADMM iter/s", nodes=srch.nodes - np0,
                      nodes_per_s=round((srch.nodes - np0) / dtp, 2),
                      iters_per_node=round((srch.iters - ip0) / max(1.0, float(srch.nodes - np0)), 1))

    # ---- extra leg (BASELINE configs[2]): the same MIQP stream with `batch_width` leaves in flight per rank
    #      (not part of `value`).  Two forms: WAVES (each wave one batched device call: a wave waits for its
    #      slowest leaf, vectors cross PCIe) and the STREAM on the device-resident leaf pool (columns refilled
    #      between chunks, 64-byte digests back: miosqp_amd/stream.py) -------------------------------------
    batched = None
    if "batched" in legs:
        waves = None  # (--batch-waves 0: the stream alone, e.g. for a kernel table of that form only)
        if args.batch_waves > 0:
            next_instance()
            run_steps(6, args.batch_width, True)  # warm-up: graph capture, allocation, frontier ramp-up
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
            if td is not None:
                tb = torch.tensor([dtb], dtype=torch.float64, device=comm.device)
                td.all_reduce(tb, op=td.ReduceOp.MAX)
                dtb = float(tb.item())
            waves = dict(waves=args.batch_waves, nodes=float(totb[1]),
                         mean_wave=round(float(totb[1]) / max(1, args.batch_waves * world), 1),
                         node_iters_per_s=round(float(totb[0]) / dtb, 1), nodes_per_s=round(float(totb[1]) / dtb, 2),
                         lockstep_iters=bit, device_us_per_lockstep_iter=round(1e3 * bms / max(1, bit), 2),
                         device_node_iters_per_s=round(bnode / max(1e-9, bms) * 1e3, 1))
            waves["end_to_end_over_device"] = round(waves["node_iters_per_s"] /
                                                    max(1.0, waves["device_node_iters_per_s"] * world), 3)
        # the stream: new MIQP, ramp-up until the columns are busy, then a timed stretch.  With more than one rank
        # every rank streams its own leaf pool and the ranks meet every few chunks (dist.ShardedStream)
        from miosqp_amd import stream as stream_mod
        next_instance()
        native = args.stream_driver == "native"
        if world == 1:
            ss = (stream_mod.NativeStreamSearch if native else stream_mod.StreamSearch)(model, columns=args.batch_width)
            sh = None
            stepper, restart = ss.step, ss.begin_instance
        else:
            sh = dist.ShardedStream(model, comm, columns=args.batch_width, exchange_every=args.stream_exchange,
                                    search=stream_mod.NativeStreamSearch(model, columns=args.batch_width) if native else None)
            ss = sh.ss
            stepper, restart = sh.step, sh.begin_instance

        class _StreamCount(object):  # what next_instance() reads: nodes of the tree that just closed, over all ranks
            base = 0
            global_nodes = property(lambda self: (sh.global_nodes if sh is not None else ss.nodes - self.base))

            def begin_instance(self):
                self.base = ss.nodes

        sc = _StreamCount()
        sc.base = ss.nodes

        def stream_steps(count):
            for _ in range(count):
                if stepper() == 0:
                    next_instance(sc)
                    restart()

        stream_steps(args.stream_warmup)
        sync()
        eng.batch_stats(reset=True)
        n2, i2, c2 = ss.nodes, ss.iters, ss.chunks
        del stream["closed"][:]
        t2 = time.perf_counter()
        stream_steps(args.stream_chunks)
        if native:
            ss.step(rounds=-1)  # the launches in flight belong to the timed region: waited for, their digests absorbed and counted
        else:
            eng.pool_collect(0)  # (Python driver: the launch in flight belongs to the timed region, its digests are not counted)
        sync()
        dts = time.perf_counter() - t2
        sms, sit, snode = eng.batch_stats()
        tots = comm.sum([ss.iters - i2, ss.nodes - n2, sms, float(sit), float(snode), float(ss.chunks - c2)])
        if td is not None:
            tb = torch.tensor([dts], dtype=torch.float64, device=comm.device)
            td.all_reduce(tb, op=td.ReduceOp.MAX)
            dts = float(tb.item())
        batched = dict(width=args.batch_width, form="stream on the device-resident leaf pool" + (", host side in the library" if native else ", host side in Python") +
                       ("" if world == 1 else ", one pool per rank, incumbent + dry-rank feed every %d chunks"
                        % args.stream_exchange),
                       chunks=ss.chunks - c2, nodes=float(tots[1]),
                       node_iters_per_s=round(float(tots[0]) / dts, 1), nodes_per_s=round(float(tots[1]) / dts, 2),
                       iters_per_node=round(float(tots[0]) / max(1.0, float(tots[1])), 1),
                       lockstep_iters=sit, device_us_per_lockstep_iter=round(1e3 * sms / max(1, sit), 2),
                       column_occupancy=round(float(tots[4]) / float(max(1.0, args.batch_width * float(tots[3]))), 3),
                       device_node_iters_per_s=round(float(tots[4]) / max(1e-9, float(tots[2]) / world) * 1e3, 1),
                       end_to_end_over_device=round(1e-3 * (float(tots[2]) / world) / dts, 3),
                       open_leaves=len(ss.open), pool_slots_free=len(ss.free), dropped_at_refill=ss.dropped,
                       waves=waves)
        closed_s = list(stream["closed"])
        bt = dict(closed_in_timed_region=len(closed_s))
        if len(closed_s) >= 2:
            gaps = [closed_s[k][0] - closed_s[k - 1][0] for k in range(1, len(closed_s))]
            bt.update(mean_ms_to_close=round(1e3 * float(np.mean(gaps)), 3),
                      mean_nodes_per_tree=round(float(np.mean([c[1] for c in closed_s[1:]])), 1),
                      trees_per_s=round(1.0 / float(np.mean(gaps)), 2))
            if trees.get("mean_nodes_per_tree"):
                # nodes the stream spends per closed tree over what the node-at-a-time search of `value` spends (other
                # instances of the same stream of MIQPs; with one rank that search is the sequential one)
                bt["node_ratio_vs_headline"] = round(bt["mean_nodes_per_tree"] / trees["mean_nodes_per_tree"], 2)
        batched["trees"] = bt
        batched["note"] = ("the compiled stream driver pushes as many leaves per round as the launch in flight has freed when "
                           "it looks: node counts (not results) vary from run to run") if native else None
        if sh is not None:
            batched["leaves_moved_rank0"] = sh.moved
        # the same tree on TWO pools of this GPU (stream.MultiPoolSearch: two engines, two host threads): the sweeps
        # of one pool fill the serial part of the other's chunk.  Reported next to the single pool, not instead of it
        if world == 1 and args.pools > 1:
            rngs = [np.random.RandomState(args.seed + 777) for _ in range(args.pools)]

            def make_model():
                # (each pool's chunk is one persistent launch that takes the whole chip: the two pools' launches take turns,
                #  a pool's test / harvest / refill runs under the other pool's sweeps; were two launches ever to split the
                #  CUs between them, both are called off within 100 ms and those engines go on with two launches per iteration)
                return setup_model(prob, dict(qs), backend=None)[0]

            def reroot(k, mdl):
                r = rngs[k]  # every pool draws the same numbers
                mdl.update_vectors(q=r.randn(cfg["n"]), l=-2 + r.rand(m_orig), u=2 + r.rand(m_orig))

            mp = stream_mod.MultiPoolSearch(make_model, pools=args.pools, columns=args.batch_width,
                                            exchange_every=args.stream_exchange, driver=args.stream_driver)
            mp.steps(args.stream_warmup, reroot)
            torch.cuda.synchronize()
            a0 = [(sh_.ss.nodes, sh_.ss.iters) for sh_ in mp.sh]
            for mdl in mp.models:
                mdl.work.solver.batch_stats(reset=True)
            t3 = time.perf_counter()
            mp.steps(args.stream_chunks, reroot)
            for sh_, mdl in zip(mp.sh, mp.models):
                if args.stream_driver == "native":
                    sh_.ss.step(rounds=-1)  # (absorbed and counted, as for the single pool)
                else:
                    mdl.work.solver.pool_collect(0)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t3
            st3 = [mdl.work.solver.batch_stats() for mdl in mp.models]
            nd = sum(sh_.ss.nodes - a[0] for sh_, a in zip(mp.sh, a0))
            it = sum(sh_.ss.iters - a[1] for sh_, a in zip(mp.sh, a0))
            batched["pools_%d" % args.pools] = dict(
                form="%d pools x %d columns on this GPU, one tree (stream.MultiPoolSearch)" % (args.pools, args.batch_width),
                node_iters_per_s=round(it / dtp, 1), nodes_per_s=round(nd / dtp, 2),
                device_us_per_lockstep_iter_per_pool=[round(1e3 * s3[0] / max(1, s3[1]), 2) for s3 in st3],
                column_occupancy=[round(s3[2] / float(max(1, args.batch_width * s3[1])), 3) for s3 in st3],
                leaves_moved=[sh_.moved for sh_ in mp.sh],
                persistent_fallbacks=[mdl.work.solver.batch_pers_fallbacks() for mdl in mp.models])
            for mdl in mp.models:
                mdl.work.solver.close()
        model.work.leaves = []  # the pool owns the open leaves of this instance
        next_instance()

    if rank == 0:
        fs = eng.factor_stats()
        kern = []
        if fs["coop"]:
            # the cooperative grid stays resident for a whole call of the search (k_coop_run: ONE launch takes node after
            # node from the host's mailbox; where it cannot, one launch of k_coop IS one node relaxation): average launch
            # from the HIP events around the launches of the timed region -- a resident launch's time includes the host's
            # turn-around between its nodes
            launches = max(1, loop_launches if loop_launches else nodes_here)
            resident = grid_resident
            us, by = 1e3 * loop_ms / launches, fs["bytes_per_iter"] * loop_iters / launches
            kern.append(dict(kernel="k_coop_run" if resident else "k_coop", usec=round(us, 3), bytes=round(by),
                             gbs=round(by / max(us, 1e-9) * 1e-3, 1), launches=launches,
                             nodes_per_launch=round(nodes_here / launches, 1),
                             usec_per_node=round(1e3 * loop_ms / max(1, nodes_here), 3),
                             iterations_per_launch=round(loop_iters / launches, 1)))
            if resident:
                # a resident launch is as long as the call of the search it serves: the launches of one process differ in
                # length (warm-up: --warmup nodes, timed region: --steps nodes), so a kernel table's AVERAGE launch says
                # nothing -- its TOTAL does: every launch of the process (warm-up + timed), their nodes and iterations,
                # and the fraction they give together
                tot_us, tot_it = 1e3 * (wu_ms + loop_ms), wu_iters + loop_iters
                kern[-1]["all_launches_of_the_process"] = dict(
                    launches=wu_launches + launches, nodes=wu_nodes + nodes_here, iterations=tot_it, usec_total=round(tot_us, 1),
                    usec_per_node=round(tot_us / max(1, wu_nodes + nodes_here), 3),
                    frac=round(fs["bytes_per_iter"] * tot_it / max(tot_us, 1e-9) * 1e-3 / HBM_PEAK_GBS, 4),
                    what="warm-up + timed region (with --legs none --no-probes nothing else launches the kernel): compare "
                         "usec_total with calls x average of the rocprofv3 kernel table of the same command")
            it_us, it_bytes = (0.0, 0.0) if args.no_probes else eng.time_kernel(4, 2000)
        elif fs["pers"]:  # likewise one launch per node
            launches = max(1, nodes_here)
            us, by = 1e3 * loop_ms / launches, fs["bytes_per_iter"] * loop_iters / launches
            kern.append(dict(kernel="k_pers_small" if fs["pers_small"] else "k_pers", usec=round(us, 3), bytes=round(by),
                             gbs=round(by / max(us, 1e-9) * 1e-3, 1), bytes_moved=round(fs["bytes_moved_per_iter"] * loop_iters / launches),
                             launches=launches, iterations_per_launch=round(loop_iters / launches, 1)))
            it_us, it_bytes = (0.0, 0.0) if args.no_probes else eng.time_kernel(4, 200)
        else:
            names = ["k_fold_fwd", "k_fold_bwd"] if fs["fold"] else KERNELS
            if args.no_probes:  # no extra launches: the whole iteration from the loop's own events
                us = 1e3 * loop_ms / max(1, loop_iters)
                kern.append(dict(kernel="+".join(names), usec=round(us, 3), bytes=fs["bytes_per_iter"],
                                 gbs=round(fs["bytes_per_iter"] / max(1e-9, us) * 1e-3, 1)))
                it_us = us
            else:
                for k, nm in enumerate(names):
                    us, by = eng.time_kernel(k, 300)
                    kern.append(dict(kernel=nm, usec=round(us, 3), bytes=by, gbs=round(by / us * 1e-3, 1)))
                it_us, it_bytes = eng.time_kernel(4, 100)
        dom = max(kern, key=lambda d: d["usec"])
        traffic, tsrc = pmc_traffic(dom["kernel"], nodes_per_launch=dom.get("nodes_per_launch"))
        # What binds the dominant kernel.  `achieved` / `frac` follow the contract: ALGORITHMIC bytes (SURVEY
        # sec. 8d's per-iteration figure x iterations per launch) over the measured launch time against the HBM
        # peak.  For the cooperative solver those bytes never leave the register file after the first
        # iteration, so the kernel is bound by its one all-to-all hand-off per iteration, not by HBM:
        # `hbm_measured_gbs` (PMC traffic / time) is what HBM really delivers.
        roof = dict(bound="exchange-latency" if fs["coop"] else "hbm",
                    roofline="hbm", kernel=dom["kernel"], achieved=dom["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(dom["gbs"] / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_source=(("committed profile %s (separate rocprofv3 --pmc passes of a node-only run of the same "
                                     "device code; not measured in this run)" % tsrc) if traffic else tsrc) if tsrc else None,
                    hbm_measured_gbs=round(traffic / dom["usec"] * 1e-3, 1) if traffic else None,
                    bytes_per_launch=dom["bytes"], usec_per_launch=dom["usec"], kernels=kern,
                    iteration=dict(bytes=fs["bytes_per_iter"],
                                   usec_in_timed_region=round(1e3 * loop_ms / max(1, loop_iters), 3),
                                   usec_back_to_back=round(it_us, 3),
                                   achieved=round(fs["bytes_per_iter"] * loop_iters / max(1e-9, loop_ms) * 1e-6, 1),
                                   frac=round(fs["bytes_per_iter"] * loop_iters / max(1e-9, loop_ms) * 1e-6 /
                                              HBM_PEAK_GBS, 4)),
                    timing="HIP events on the engine's stream",
                    note="the factor is register-resident in this form (explicit KKT inverse spread over the CUs): "
                         "`achieved` counts algorithmic bytes, `hbm_measured_gbs` what HBM delivered; the "
                         "HBM-streaming form of the same workload is the `stream` leg" if fs["coop"] else None)
        out = dict(metric="ADMM iterations/s (random_miqp n=%d m=%d p=%d, node-at-a-time B&B)" %
                          (cfg["n"], cfg["m"], cfg["p"]),
                   value=round(iters / dt_max, 1), unit="ADMM iter/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=round(1e3 * dt_max / args.steps, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   nodes_per_s=round(nodes / dt_max, 2), iters_per_node=round(iters / max(1.0, nodes), 1),
                   nodes=nodes, trees=trees,
                   config=dict(workload="%s: random_miqp n=%d m=%d p=%d density %.2f seed %d, "
                                        "%s per rank per step, leaves sharded over %d GPU(s)%s" %
                                        ({"cfg1": "BASELINE configs[0]'s shape on the GPU", "cfg2": "BASELINE configs[1]",
                                          "cfg5": "BASELINE configs[4]",
                                          "cfg5x": "not a BASELINE config (configs[4]'s shape beyond the Infinity Cache)"}
                                         [args.config],
                                         cfg["n"], cfg["m"], cfg["p"], cfg["density"], args.seed,
                                         ("node relaxations for %.1f ms" % args.step_budget_ms) if budget
                                         else "%d node(s)" % args.wave, world,
                                         "" if rpt >= world else " -- %d ranks per tree (incumbent exchange and leaf hand-over "
                                         "inside the group), %d trees of the MIQP stream at a time" % (rpt, world // rpt)),
                               instance=problems.instance_digest(prob), nnz_L=fs["nnz_L"],
                               factor_form=("explicit KKT inverse in registers, cooperative grid resident over the nodes of a "
                                            "search_run call (k_coop_run)" if grid_resident else
                                            "explicit KKT inverse in registers, cooperative launch per node")
                               if fs["coop"] else ("product form L^-1" if fs["fold"] else "L") + ", persistent streaming launch per node"
                               if fs["pers"] else "product form L^-1 (2 launches/iteration)" if fs["fold"]
                               else "L (4 launches/iteration)",
                               qp_settings=problems.QP_SETTINGS, rho=0.1, setup_s=round(t_setup, 3),
                               coop_fallbacks=fs["coop_fallbacks"], replicated_resyncs=srch.resyncs,
                               host_loop="C++ host library on device-resident leaves (miosqp_qp_search_*)" if hosted
                               else "Python (bnb.Workspace driving miosqp_qp_solve_node)",
                               comm=type(comm).__name__ + ("/nccl" if (td is not None and not one_dev) else "")),
                   roofline=roof)
        out["config"]["instances_in_timed_region"] = instances
        if node_us is not None and node_us[3] > 0:
            # the timed region is a few dozen nodes on the driver's command: the spread over the nodes travels with the mean
            out["usec_per_iter_over_nodes"] = dict(min=round(node_us[0], 4), median=round(node_us[1], 4), max=round(node_us[2], 4),
                                                   nodes=node_us[3], mean=round(1e3 * loop_ms / max(1, loop_iters), 4),
                                                   what="device time of a node (its launch by HIP events, or -- resident grid -- "
                                                        "the first tester's wall clock from the node's mail to its record) / "
                                                        "its ADMM iterations")
        if per_rank is not None:
            # the sharded search rank by rank: every rank appears once (the collective reached all of them), what it solved,
            # and in how many of its steps it had no leaf to solve
            out["ranks"] = [dict(rank=int(r[0]), nodes=int(r[1]), iters=int(r[2]), steps=int(r[3]), idle_steps=int(r[4]),
                                 idle_frac=round(r[4] / max(1.0, r[3]), 3), leaves_given=int(r[5]), seconds=round(float(r[6]), 4),
                                 tree_group=int(r[7]))
                            for r in per_rank]
        if pyloop is not None:
            out["python_loop"] = pyloop
        if batched is not None and batched["lockstep_iters"] > 0:
            bk = []
            if not args.no_probes:
                bnames = ["kbm_fwd", "kbm_bwd"] if fs["fold"] else ["kb_" + k[2:] for k in KERNELS]
                for k, nm in enumerate(bnames):
                    us, by = eng.time_kernel(10 + k, 30)
                    bk.append(dict(kernel=nm, usec=round(us, 2), bytes=by, gbs=round(by / us * 1e-3, 1)))
            batched["kernels_full_width_back_to_back"] = bk
            batched["sweeps"] = ("one persistent launch per chunk of check_termination iterations, factor in registers and LDS "
                                 "(kbp1)" if fs.get("batch_pers") else "two launches per lock-step iteration (kbm_fwd, kbm_bwd)")
            batched["persistent_fallbacks"] = eng.batch_pers_fallbacks()
            if fs.get("batch_pers") and not args.no_probes:
                us, _ = eng.time_kernel(15, 100)
                batched["persistent_sweeps_us_per_lockstep_iter"] = round(us, 2)
            # SURVEY 8d's config-3 accounting: matrix terms once per lock-step iteration, vector terms per column;
            # flops: both sweeps, 2 flop per factor entry (product form: n M + n (n - 1) / 2 entries per sweep)
            nn, MM = cfg["n"], cfg["m"] + cfg["p"]
            per_col = (6 * nn + 16 * MM) * 8
            alg = (fs["bytes_per_iter"] - per_col) + per_col * args.batch_width
            flops = 4.0 * (nn * MM + nn * (nn - 1) / 2) * args.batch_width
            us = batched["device_us_per_lockstep_iter"]
            batched["roofline"] = dict(bound="fp64 matrix cores + L2->L1 operand traffic (two dense fp64 GEMMs per iteration, "
                                             "16 x 32 tiles; the factor is read from L2 / Infinity Cache, not HBM)",
                                       algorithmic_bytes_per_lockstep_iter=int(alg), achieved_gbs=round(alg / us * 1e-3, 1),
                                       frac_of_hbm_peak=round(alg / us * 1e-3 / HBM_PEAK_GBS, 4),
                                       flops_per_lockstep_iter=int(flops), tflops=round(flops / us * 1e-6, 2),
                                       fp64_mfma_peak_tflops=78.6, frac_of_fp64_peak=round(flops / us * 1e-6 / 78.6, 4),
                                       note="device time per lock-step iteration includes the per-chunk termination test, "
                                            "refill and harvest")
            out["batched"] = batched
        if world == 1 and args.config == "cfg2":
            if "stream" in legs:
                out["stream"] = stream_leg(prob, local_rank)
            if "config5" in legs:
                out["config5"] = large_leg(args.seed, local_rank)
            if "config4" in legs:
                out["config4"] = mpc_leg(local_rank)
            if "config1" in legs:
                out["config1"] = small_leg(args.seed, local_rank)
            if "rho_auto" in legs and hosted:
                out["rho_auto"] = rho_auto_leg(prob, cfg, args.seed, local_rank, max(150, args.steps),
                                               batched_chunks=args.stream_chunks if "batched" in legs else 0)
                # both settings of rho side by side at the top level: "ADMM iterations/s" rewards the setting that needs more
                # iterations per node, nodes/s and the time to close a tree -- the other half of BASELINE's metric -- do not
                ra = out["rho_auto"]
                t01 = ra["same_miqp"]["rho_default"]
                out["by_rho"] = {
                    "same_miqp": dict(instance=problems.instance_digest(prob), upper_glob_agree=ra["same_miqp"]["upper_glob_agree"],
                                      upper_glob={"0.1": t01["upper_glob"], "auto": ra["one_tree"]["upper_glob"]},
                                      what="ms_to_close_tree / nodes_per_tree of both settings are of this one MIQP, closed from "
                                           "its root; the rates are of the MIQP stream of the timed region"),
                    "0.1 (frozen default: `value`)": dict(
                        rho=0.1, admm_iter_per_s=out["value"], nodes_per_s=out["nodes_per_s"], iters_per_node=out["iters_per_node"],
                        ms_to_close_tree=t01["ms_to_close"], nodes_per_tree=t01["nodes"]),
                    "auto (chosen once at set-up)": dict(
                        rho=ra["rho"], admm_iter_per_s=ra["value"], nodes_per_s=ra["nodes_per_s"], iters_per_node=ra["iters_per_node"],
                        ms_to_close_tree=ra["one_tree"]["ms_to_close"], nodes_per_tree=ra["one_tree"]["nodes"],
                        setup_s=ra["setup_s"])}
        if world == 1 and "cpu" in legs:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_seconds)
        print(json.dumps(out))
    if td is not None:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
