#!/usr/bin/env python3
"""bench.py -- Lambda-solve throughput of the MI355X path on a Venice-871-shaped synthetic BA.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload venice871] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one numeric Lambda-solve of the hot path (reference: CNonlinearSolver_Lambda::Optimize's
linear solve, include/slam/NonlinearSolver_Lambda.h:605-626, through CLinearSolver_Schur
include/slam/LinearSolver_Schur.h:1623-1935): landmark elimination (C^-1, W = -U C^-1, S = A + W U^T,
reduced rhs), dense Cholesky of the reduced camera system, triangular solves, landmark
back-substitution. Lambda and eta are assembled ON THE DEVICE by the HIP assembly kernels before the
timed region and stay resident in HBM; symbolic analysis (ordering, plans) is done once and reported
separately, exactly as the reference amortizes it (metric 1).

value = nnzb(Lambda upper incl. diagonal) x steps / wall time  [block-nnz/s], whole job.
N > 1: landmarks are sharded over the ranks, every rank forms its partial Schur
complement, packs the upper trapezoid of S | rhs, ONE RCCL all-reduce (torch.distributed "nccl")
sums it, every rank factors S and back-substitutes its own landmarks.
  --scaling weak (default): every rank brings its own Venice-sized landmark shard (530 304 points,
    2 838 740 observations) seen by the same 871 cameras: the job is ONE bundle adjustment with
    N x 530 304 landmarks, value = sum of the shards' block-nnz / time. Per-GPU work is fixed.
  --scaling strong: the single Venice problem, landmarks dealt round-robin over the ranks. The dense
    factor of the 5226^2 reduced system is replicated, so this mode is bounded by it.

Extra fields: gn_iters_per_s (assembly + solve = one Gauss-Newton/LM iteration of device work),
phase_ms, analyze_s, roofline (dominant kernel = MFMA f64 trailing update of the dense factor),
cpu_baseline (the REFERENCE's own CLinearSolver_Schur compiled from /root/reference into
oracle/_ref, timed on this box's host cores on the same Lambda).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "Λ-solve block-nnz/sec + batch GN iters/sec, Venice BA, 1→8 MI355X"
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak (vendor datasheet; MI355X_MICROARCH.md lists no fp64 row)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="venice871")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-solves", type=int, default=2)
    ap.add_argument("--cpu-backend", default="auto", choices=["auto", "schur", "cholmod", "csparse", "uberblock"],
                    help="reference backend of the cpu_baseline leg. auto = the reference's FASTEST path for the workload "
                         "(CLinearSolver_Schur + dense LLT for BA, UberBlock for pose graphs). north_star's CHOLMOD path "
                         "takes ~150 s per Venice-shaped solve (DESIGN.md 5): run it explicitly with --cpu-backend cholmod --cpu-solves 1")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank brings its own Venice-sized landmark set seen by the same 871 "
                         "cameras (per-GPU work fixed); strong = the one Venice problem sharded by landmarks")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a single GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    import torch
    import torch.distributed as dist
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from slam_plus_plus_amd import api, synth

    def barrier_sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- problem (same seed on every rank) and device-resident inputs
    t0 = time.time()
    weak = world > 1 and args.scaling == "weak"
    if weak and args.workload == "venice871":
        # same cameras (deterministic circle), a different landmark / observation set per rank
        prob = synth.ba_problem(871, 530304, 2838740, 871 + rank, heavy_tail=True, name="venice871")
    else:
        prob = synth.make(args.workload)
    gen_s = time.time() - t0
    ctx = api.Context(local_rank, api.FLAG_PROFILE)
    st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
    d_in = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
    d_vals = api.DeviceArray(ctx, st.nvals)
    d_eta = api.DeviceArray(ctx, st.n)
    d_rhs = api.DeviceArray(ctx, st.n)

    def assemble():
        ctx.assemble_device(d_in[0].ptr, d_in[1].ptr, d_in[2].ptr, d_in[3].ptr, prob.damping, d_vals.ptr, d_eta.ptr)

    assemble()
    ctx.synchronize()
    t0 = time.time()
    if weak:
        ctx.set_shard(0, 1)       # this rank owns ALL landmarks of its own problem (and its share of A / eta)
    else:
        ctx.set_shard(rank, world)
    ctx.analyze(st, api.MODE_AUTO)
    analyze_s = time.time() - t0
    mode = ctx.info("MODE")
    nnzb = st.nnzb

    S = P = None
    schur = mode in (api.MODE_SCHUR, api.MODE_SCHUR_SPARSE)
    if schur and world > 1:
        S = torch.empty(ctx.schur_buffer_size(), dtype=torch.float64, device="cuda")
        P = torch.empty(ctx.schur_packed_size(), dtype=torch.float64, device="cuda")  # upper trapezoid only

    def solve():
        d_rhs.copy_from(d_eta)
        if S is None:
            code = ctx.factor_solve_device(d_vals.ptr, d_rhs.ptr)
        else:
            ctx.schur_form(d_vals.ptr, d_rhs.ptr, S.data_ptr())
            ctx.schur_pack(S.data_ptr(), P.data_ptr())
            ctx.synchronize()
            dist.all_reduce(P)            # the one data-path collective: reduced camera system + reduced rhs
            torch.cuda.synchronize()
            ctx.schur_unpack(P.data_ptr(), S.data_ptr())
            code = ctx.schur_finish(d_vals.ptr, S.data_ptr(), d_rhs.ptr)
        if code != 0:
            raise SystemExit("factorization failed (not positive definite): code %d" % code)

    # ---- timed region: exactly K Lambda-solves, profiling instrumentation OFF (the hipEvents around
    # the phases and around every trailing-update launch cost a barrier packet each)
    ctx.set_profiling(False)
    for _ in range(args.warmup):
        solve()
    ctx.synchronize()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        solve()
        if os.environ.get("BENCH_DEBUG"):
            print("step %.3f ms" % (1e3 * (time.perf_counter() - t1)), file=sys.stderr)
    ctx.synchronize()
    barrier_sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # ---- separate profiled pass: phase breakdown and the hipEvent-timed dominant kernel (last solve)
    ctx.set_profiling(True)
    for _ in range(3):
        solve()
    ctx.synchronize()
    phase = ctx.phase_ms()
    dom_ms, dom_n, dom_flops = ctx.dominant_kernel()
    ctx.set_profiling(False)

    x_solve = d_rhs.download()   # solution of the timed Lambda-solves (the later loops reuse the buffers)

    # ---- second loop: full GN iteration of device work (assembly + eta + solve)
    gn_steps = max(3, args.steps // 2)
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(gn_steps):
        assemble()
        solve()
    ctx.synchronize()
    barrier_sync()
    dt_gn = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt_gn], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_gn = float(tt.item())
    ctx.set_profiling(True)
    assemble()
    ctx.synchronize()
    assemble_ms = ctx.phase_ms()["assemble"]

    # ---- third loop (BA, one GPU): the whole damped Gauss-Newton iteration in HBM -- device linearization
    # in the reference's parameterization (spp_ba_linearize_device), assembly, solve, ||dx|| and the
    # vertex update (spp_ba_update_device); the state is reset from a device copy every iteration so that
    # all iterations do the same work. Host traffic per iteration: 8 bytes.
    resident = None
    if world == 1 and "geometry" in prob and prob.geometry.get("kind") in ("se2", "se3"):
        # pose graphs: spp_se2_/se3_linearize_device + assembly + sparse solve + ||dx|| + vertex update
        pg = synth.pose_graph_states(prob)
        dof, nv_, ne_ = pg["dof"], pg["poses"].shape[0], pg["v0"].size
        d_pg = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(pg[k]).ravel()) for k in ("poses", "meas", "v0", "v1")}
        d_pw = api.DeviceArray(ctx, pg["poses"].size)
        r_J0, r_J1, r_r = api.DeviceArray(ctx, d_in[0].n), api.DeviceArray(ctx, d_in[1].n), api.DeviceArray(ctx, d_in[3].n)
        r_vals, r_eta = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n)
        lin = ctx.se2_linearize_device if dof == 3 else ctx.se3_linearize_device
        upd = ctx.se2_update_device if dof == 3 else ctx.se3_update_device

        def gn_resident():
            d_pw.copy_from(d_pg["poses"])
            lin(ne_, d_pg["v0"].ptr, d_pg["v1"].ptr, d_pw.ptr, d_pg["meas"].ptr, r_J0.ptr, r_J1.ptr, r_r.ptr)
            ctx.assemble_device(r_J0.ptr, r_J1.ptr, d_in[2].ptr, r_r.ptr, 0.0, r_vals.ptr, r_eta.ptr)
            if ctx.factor_solve_device(r_vals.ptr, r_eta.ptr) != 0:
                raise SystemExit("resident GN: factorization failed")
            return upd(nv_, d_pw.ptr, r_eta.ptr, apply=True)

        ctx.set_profiling(False)
        for _ in range(2):
            gn_resident()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(gn_steps):
            t1 = time.perf_counter()
            dxn = gn_resident()
            if os.environ.get("BENCH_DEBUG"):
                print("resident iteration %.3f ms" % (1e3 * (time.perf_counter() - t1)), file=sys.stderr)
        ctx.synchronize()
        dt_res = time.perf_counter() - t0
        resident = {"iters_per_s": gn_steps / dt_res, "ms_per_iter": 1e3 * dt_res / gn_steps, "dx_norm": dxn,
                    "batch_of_5_iterations_ms": 5e3 * dt_res / gn_steps,
                    "what": "device linearization (CEdgePose%dD, analytic) + assembly + sparse multifrontal solve + ||dx|| + "
                            "vertex update, everything resident in HBM; state reset by one device copy per iteration (included)" % (2 if dof == 3 else 3)}
    elif world == 1 and "geometry" in prob:
        sc = synth.ba_states(prob)
        d_s = {k: api.DeviceArray.from_host(ctx, np.ascontiguousarray(v).ravel()) for k, v in sc.items()}
        d_cw, d_pw = api.DeviceArray(ctx, sc["cams"].size), api.DeviceArray(ctx, sc["points"].size)
        no_, nc_, np_ = prob.v0.size, sc["cams"].shape[0], sc["points"].shape[0]
        # own buffers: the Lambda of the timed solves stays intact for the CPU baseline comparison below
        r_J0, r_J1, r_r = api.DeviceArray(ctx, d_in[0].n), api.DeviceArray(ctx, d_in[1].n), api.DeviceArray(ctx, d_in[3].n)
        r_vals, r_eta = api.DeviceArray(ctx, st.nvals), api.DeviceArray(ctx, st.n)

        def gn_resident():
            d_cw.copy_from(d_s["cams"])
            d_pw.copy_from(d_s["points"])
            ctx.ba_linearize_device(no_, d_s["cam_of"].ptr, d_s["pt_of"].ptr, d_cw.ptr, d_s["intr"].ptr, d_pw.ptr,
                                    d_s["meas"].ptr, r_J0.ptr, r_J1.ptr, r_r.ptr)
            ctx.assemble_device(r_J0.ptr, r_J1.ptr, d_in[2].ptr, r_r.ptr, prob.damping, r_vals.ptr, r_eta.ptr)
            if ctx.factor_solve_device(r_vals.ptr, r_eta.ptr) != 0:
                raise SystemExit("resident GN: factorization failed")
            return ctx.ba_update_device(nc_, d_cw.ptr, d_s["cam_dxoff"].ptr, np_, d_pw.ptr, d_s["pt_dxoff"].ptr,
                                        r_eta.ptr, st.n, apply=True)

        ctx.set_profiling(False)
        for _ in range(2):
            gn_resident()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(gn_steps):
            t1 = time.perf_counter()
            dxn = gn_resident()
            if os.environ.get("BENCH_DEBUG"):
                print("resident iteration %.3f ms" % (1e3 * (time.perf_counter() - t1)), file=sys.stderr)
        ctx.synchronize()
        dt_res = time.perf_counter() - t0
        # the linearization kernel alone (hipEvent-free: wall clock around a synchronized burst)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.ba_linearize_device(no_, d_s["cam_of"].ptr, d_s["pt_of"].ptr, d_cw.ptr, d_s["intr"].ptr, d_pw.ptr,
                                    d_s["meas"].ptr, r_J0.ptr, r_J1.ptr, r_r.ptr)
        ctx.synchronize()
        lin_ms = 1e3 * (time.perf_counter() - t0) / 5
        resident = {"iters_per_s": gn_steps / dt_res, "ms_per_iter": 1e3 * dt_res / gn_steps, "linearize_ms": lin_ms,
                    "linearize_gbs": (no_ * (160 + 16 + 8) + 0.0) / (lin_ms * 1e-3) * 1e-9, "dx_norm": dxn,
                    "what": "device linearization (CEdgeP2C3D, analytic) + assembly + Schur solve + ||dx|| + vertex update; "
                            "state reset by two device copies per iteration (included)"}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    x = x_solve
    total_nnzb = nnzb * world if weak else nnzb
    out = {
        "metric": METRIC, "value": total_nnzb * args.steps / dt, "unit": "block-nnz/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic BA (%d cams, %d points, %d observations, LM-damped)" % (
            args.workload, prob.get("nc", 0), prob.get("npts", 0), prob.v0.size) if "nc" in prob else args.workload,
            "nnzb": int(nnzb), "n": int(st.n), "mode": {api.MODE_SCHUR: "schur+dense", api.MODE_SCHUR_SPARSE: "schur+sparse reduced system",
                                                            api.MODE_SPARSE: "sparse multifrontal"}[mode],
            "n_reduced": int(ctx.info("N_REDUCED")), "schur_pairs": int(ctx.info("SCHUR_PAIRS")),
            "parallelism": ("%s: landmark shards x%d + one RCCL all-reduce of the packed reduced camera system (%.0f MB)" % (
                "weak (871 cameras, 530304 landmarks per GPU)" if weak else "strong (one Venice problem)", world,
                8e-6 * ctx.schur_packed_size())) if (world > 1 and schur) else "single GPU"},
        "gn_iters_per_s": gn_steps / dt_gn, "ms_per_gn_iter": 1e3 * dt_gn / gn_steps, "assemble_ms": assemble_ms,
        "phase_ms": {k: round(v, 4) for k, v in phase.items()},
        "phase_ms_note": "separate pass with SPP_FLAG_PROFILE on (hipEvents per phase and per trailing-update launch); ms_per_step is timed with profiling off", "analyze_s": round(analyze_s, 3),
        "generate_s": round(gen_s, 2), "solution_norm": float(np.linalg.norm(x)),
    }
    if resident is not None:
        out["gn_resident"] = resident
    # ---- roofline of the dominant kernel (hipEvents on the ctx stream, last timed solve)
    if dom_n > 0 and dom_ms > 0:
        achieved = dom_flops / (dom_ms * 1e-3) * 1e-12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": "spp::gemm_tn_mixed_kernel (MFMA f64 16x16x4 trailing update of the dense factor: 128x128 tiles, the tail of every launch cut into 64x64 quarters)",
                           "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                           "launches_per_solve": int(dom_n), "avg_launch_ms": dom_ms / dom_n,
                           "flops_per_launch": dom_flops / dom_n,
                           "standalone_update_tflops": (lambda m: (128.0 * m * (m + 1) + 256.0 * m) / ctx.microbench_update(m, 5) * 1e-9)(5120),
                           "measured_mfma_f64_peak": ctx.microbench_mfma_f64(4000),
                           "measured_copy_gbs": ctx.microbench_copy(1 << 30, 10),
                           "measured_ctile_rw_gbs": ctx.microbench_ctile(8192, 5)}
        # the HBM-bound part of the solve, for reference: algorithmic bytes / time of the Schur phases
        sch_ms = phase["schur_inv"] + phase["schur_gemm"] + phase["schur_rhs"] + phase["backsubst"]
        if sch_ms > 0:
            nobs, npair = ctx.info("N_OBS"), ctx.info("SCHUR_PAIRS")
            sch_bytes = 144 * nobs * 4 + 288 * npair + 72 * ctx.info("N_LANDMARKS") + 288 * ctx.info("N_POSES") ** 2 / 2
            out["schur_hbm"] = {"achieved": sch_bytes / (sch_ms * 1e-3) * 1e-9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": sch_bytes / (sch_ms * 1e-3) * 1e-9 / PEAK_HBM_GBS}
    # ---- CPU baseline: the reference itself (oracle/_ref), same Lambda, bounded sample
    if not args.no_cpu_baseline and world == 1 and ctx.info("N_REDUCED") <= 8192:  # the reference's Schur solve is dense: 60000^2 does not fit its path
        try:
            from oracle import spp_oracle as orc
            if orc.have_ref():
                lam = st.with_vals(d_vals.download())
                eta = d_eta.download()
                backend = ("schur" if schur else "uberblock") if args.cpu_backend == "auto" else args.cpu_backend
                rs = orc.RefSolver(backend, lam)
                secs = []
                for _ in range(args.cpu_solves):
                    code, xr, sec = rs.solve(lam.vals, eta)
                    secs.append(sec)
                out["cpu_baseline"] = {"value": nnzb / min(secs), "unit": "block-nnz/s", "cores": 1, "kind": "reference",
                                       "sample": "%d full Lambda-solves of the same %s system by the reference's %s "
                                                 "(symbolic included in the first), best of %d: %.2f s" % (
                                                     args.cpu_solves, args.workload,
                                                     {"schur": "CLinearSolver_Schur + dense Eigen LLT", "uberblock": "CLinearSolver_UberBlock",
                                                      "cholmod": "CLinearSolver_CholMod", "csparse": "CLinearSolver_CSparse"}[backend],
                                                     args.cpu_solves, min(secs)),
                                       "rel_diff_gpu_vs_reference": float(np.linalg.norm(x - xr) / np.linalg.norm(xr))}
            else:
                out["cpu_baseline"] = {"value": None, "unit": "block-nnz/s", "cores": 1, "kind": "reference",
                                       "sample": "oracle/_ref/libspp_ref.so not present"}
        except Exception as e:  # the baseline must never take the measurement down
            out["cpu_baseline"] = {"value": None, "unit": "block-nnz/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
