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
sums it, every rank factors S and back-substitutes its own landmarks. All of it is enqueued on ONE stream
(the ctx adopts torch's current stream: no host synchronization around the collective).
  --scaling strong (default = BASELINE config 4 / the metric): the ONE Venice problem, landmarks dealt
    round-robin over the ranks; `value` is its block-nnz / time. The dense factor of the 5226^2 reduced
    system and the collective are replicated work: the Amdahl bound is stated in config.parallelism.
  the same run then times the weak-scaling variant as an EXTRA field (`weak_scaling`): every rank brings
    its own Venice-sized landmark shard (530 304 points, 2 838 740 observations) seen by the same 871
    cameras -- one bundle adjustment with N x 530 304 landmarks, per-GPU work fixed. --scaling weak makes
    that variant the headline instead (labelled so).

Extra fields: gn_iters_per_s (assembly + solve = one Gauss-Newton/LM iteration of device work),
phase_ms, analyze_s, roofline (dominant kernel = MFMA f64 trailing update of the dense factor),
cpu_baseline (the REFERENCE's own solvers compiled from /root/reference into oracle/_ref, timed on this box's
host cores: headline = its fastest path on the same Lambda (CLinearSolver_Schur + dense LLT for BA), `cholmod` =
north_star's CLinearSolver_CholMod on a bounded sample), dropin_ms (the host-pointer entry of the boundary).
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
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default, BASELINE config 4) = the one Venice problem sharded by landmarks; "
                         "weak = every rank brings its own Venice-sized landmark set seen by the same 871 cameras")
    ap.add_argument("--graph", default=None, help="take the workload from a graph file (EDGE_SE2 / EDGE3[:AXISANGLE] / "
                    "VERTEX_CAM + VERTEX_XYZ + EDGE_PROJECT_P2MC) instead of the synthetic generator")
    ap.add_argument("--cpu-cholmod", default="auto", choices=["auto", "sample", "full", "off"],
                    help="north_star's CPU path (CLinearSolver_CholMod) timed in this run: on the whole Lambda when that takes "
                         "seconds (pose graphs, Ladybug), on a quarter of the landmarks for the Venice shape (~12 s; the whole "
                         "Lambda takes ~150 s: --cpu-cholmod full)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a single GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--collective", default="direct", choices=["direct", "allreduce"],
                    help="N > 1: direct = reduce-scatter by one all-to-all over the point-to-point xGMI links + local sum + "
                         "all-gather; allreduce = one RCCL all_reduce (ring)")
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
    graph_note = None
    if args.graph:
        from slam_plus_plus_amd import formats
        prob, graph_note = formats.problem_from_graph(args.graph)
        if "nc" in prob:  # BA input: the reference switches to Levenberg-Marquardt (src/slam_app/Main.cpp:203-208)
            h = max(float(np.square(prob.J0).reshape(prob.v0.size, 6, 2).sum(axis=2).max()),
                    float(np.square(prob.J1).reshape(prob.v0.size, 3, 2).sum(axis=2).max()))
            prob["damping"] = 1e-3 * h
        args.workload = os.path.basename(args.graph)
    elif weak and args.workload == "venice871":
        # same cameras (deterministic circle), a different landmark / observation set per rank
        prob = synth.ba_problem(871, 530304, 2838740, 871 + rank, heavy_tail=True, name="venice871")
    else:
        prob = synth.make(args.workload)
    gen_s = time.time() - t0

    # N > 1 over RCCL: everything runs on ONE non-blocking torch stream (the legacy default stream would serialize the
    # ctx's internal streams against it); BENCH_FORCE_TORCH_STREAM=1 rehearses that stream set-up on a single GPU
    use_torch_stream = (world > 1 and args.backend == "nccl") or bool(os.environ.get("BENCH_FORCE_TORCH_STREAM"))
    if use_torch_stream:
        torch.cuda.set_stream(torch.cuda.Stream())

    # N > 1: how the packed partial systems are summed (see Job.reduce_packed)
    collective = ["direct" if (world > 1 and args.collective == "direct") else "allreduce"]

    class Job:
        """one Lambda resident in HBM + everything a numeric solve needs; shard = (rank, world) of the landmark sharding"""
        def __init__(self, prob, shard, flags):
            self.prob = prob
            self.ctx = ctx = api.Context(local_rank, flags)
            if use_torch_stream:
                # the ctx works on torch's current stream: the RCCL all-reduce is ordered behind the pack kernel and in
                # front of the unpack kernel by the stream (torch orders its collective stream against the current one)
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            self.st = st = ctx.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
            self.d_in = [api.DeviceArray.from_host(ctx, a.ravel()) for a in (prob.J0, prob.J1, prob.Om, prob.r)]
            self.d_vals = api.DeviceArray(ctx, st.nvals)
            self.d_eta = api.DeviceArray(ctx, st.n)
            self.d_rhs = api.DeviceArray(ctx, st.n)
            self.assemble()
            ctx.synchronize()
            t0 = time.time()
            ctx.set_shard(*shard)
            ctx.analyze(st, api.MODE_AUTO)
            self.analyze_s = time.time() - t0
            self.mode = ctx.info("MODE")
            self.schur = self.mode in (api.MODE_SCHUR, api.MODE_SCHUR_SPARSE)
            self.S = self.P = None
            if self.schur and world > 1:
                self.S = torch.empty(ctx.schur_buffer_size(), dtype=torch.float64, device="cuda")
                from slam_plus_plus_amd.exchange import PackedExchange
                self.xch = PackedExchange(ctx.schur_packed_size(), world, "cuda", collective[0])
                self.P = self.xch.buf  # upper trapezoid only (+ padding to equal slices)

        def reduce_packed(self):
            """sum of the ranks' packed partial systems, the result on every rank (slam_plus_plus_amd/exchange.py)"""
            self.xch.sum()

        def assemble(self):
            d = self.d_in
            self.ctx.assemble_device(d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, self.prob.damping, self.d_vals.ptr, self.d_eta.ptr)

        def solve(self):
            ctx = self.ctx
            self.d_rhs.copy_from(self.d_eta)
            if self.S is None:
                code = ctx.factor_solve_device(self.d_vals.ptr, self.d_rhs.ptr)
            else:
                ctx.schur_form(self.d_vals.ptr, self.d_rhs.ptr, self.S.data_ptr())
                ctx.schur_pack(self.S.data_ptr(), self.P.data_ptr())
                if not use_torch_stream:
                    ctx.synchronize()
                self.reduce_packed()          # the one data-path exchange: reduced camera system + reduced rhs
                if not use_torch_stream:
                    torch.cuda.synchronize()
                ctx.schur_unpack(self.P.data_ptr(), self.S.data_ptr())
                code = ctx.schur_finish(self.d_vals.ptr, self.S.data_ptr(), self.d_rhs.ptr)
            if code != 0:
                raise SystemExit("factorization failed (not positive definite): code %d" % code)

        def timed(self, steps, warmup, debug_label=None):
            """W untimed + exactly K timed Lambda-solves, barrier + synchronize on both sides, MAX over ranks [s]"""
            self.ctx.set_profiling(False)
            for _ in range(warmup):
                self.solve()
            self.ctx.synchronize()
            barrier_sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                t1 = time.perf_counter()
                self.solve()
                if debug_label and os.environ.get("BENCH_DEBUG"):
                    print("%s %.3f ms" % (debug_label, 1e3 * (time.perf_counter() - t1)), file=sys.stderr)
            self.ctx.synchronize()
            barrier_sync()
            dt = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

    job = Job(prob, (0, 1) if weak else (rank, world), api.FLAG_PROFILE)
    ctx, st, mode, schur, analyze_s = job.ctx, job.st, job.mode, job.schur, job.analyze_s
    d_in, d_vals, d_eta, d_rhs = job.d_in, job.d_vals, job.d_eta, job.d_rhs
    assemble, solve = job.assemble, job.solve
    nnzb = st.nnzb

    # ---- timed region: exactly K Lambda-solves, profiling instrumentation OFF (the hipEvents around
    # the phases and around every trailing-update launch cost a barrier packet each)
    dt = job.timed(args.steps, args.warmup, "step")
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
    # builds the device-resident Gauss-Newton / LM iteration for a ctx: returns (iterate(reset), what, linearize-only fn or None)
    host_state = None
    if world == 1 and "geometry" in prob:  # the graph's state in host memory (generated once, outside every clock)
        host_state = synth.pose_graph_states(prob) if prob.geometry.get("kind") in ("se2", "se3") else synth.ba_states(prob)

    def build_resident(ctx_, st_):
        d_Om = api.DeviceArray.from_host(ctx_, prob.Om.ravel())
        r_J0, r_J1, r_r = api.DeviceArray(ctx_, prob.J0.size), api.DeviceArray(ctx_, prob.J1.size), api.DeviceArray(ctx_, prob.r.size)
        r_vals, r_eta = api.DeviceArray(ctx_, st_.nvals), api.DeviceArray(ctx_, st_.n)
        if prob.geometry.get("kind") in ("se2", "se3"):
            # pose graphs: spp_se2_/se3_linearize_device + assembly + sparse solve + ||dx|| + vertex update
            pg = host_state
            dof, nv_, ne_ = pg["dof"], pg["poses"].shape[0], pg["v0"].size
            d_pg = {k: api.DeviceArray.from_host(ctx_, np.ascontiguousarray(pg[k]).ravel()) for k in ("poses", "meas", "v0", "v1")}
            d_pw = api.DeviceArray(ctx_, pg["poses"].size)
            lin = ctx_.se2_linearize_device if dof == 3 else ctx_.se3_linearize_device
            upd = ctx_.se2_update_device if dof == 3 else ctx_.se3_update_device

            def iterate(reset=True):
                if reset:
                    d_pw.copy_from(d_pg["poses"])
                lin(ne_, d_pg["v0"].ptr, d_pg["v1"].ptr, d_pw.ptr, d_pg["meas"].ptr, r_J0.ptr, r_J1.ptr, r_r.ptr)
                ctx_.assemble_device(r_J0.ptr, r_J1.ptr, d_Om.ptr, r_r.ptr, 0.0, r_vals.ptr, r_eta.ptr)
                if ctx_.factor_solve_device(r_vals.ptr, r_eta.ptr) != 0:
                    raise SystemExit("resident GN: factorization failed")
                return upd(nv_, d_pw.ptr, r_eta.ptr, apply=True)
            what = ("device linearization (CEdgePose%dD, analytic) + assembly + sparse multifrontal solve + ||dx|| + vertex update, "
                    "everything resident in HBM; state reset by one device copy per iteration (included)" % (2 if dof == 3 else 3))
            return iterate, what, None
        sc = host_state
        d_s = {k: api.DeviceArray.from_host(ctx_, np.ascontiguousarray(v).ravel()) for k, v in sc.items()}
        d_cw, d_pw = api.DeviceArray(ctx_, sc["cams"].size), api.DeviceArray(ctx_, sc["points"].size)
        no_, nc_, np_ = prob.v0.size, sc["cams"].shape[0], sc["points"].shape[0]

        def linearize():
            ctx_.ba_linearize_device(no_, d_s["cam_of"].ptr, d_s["pt_of"].ptr, d_cw.ptr, d_s["intr"].ptr, d_pw.ptr,
                                     d_s["meas"].ptr, r_J0.ptr, r_J1.ptr, r_r.ptr)

        def iterate(reset=True):
            if reset:
                d_cw.copy_from(d_s["cams"])
                d_pw.copy_from(d_s["points"])
            linearize()
            ctx_.assemble_device(r_J0.ptr, r_J1.ptr, d_Om.ptr, r_r.ptr, prob.damping, r_vals.ptr, r_eta.ptr)
            if ctx_.factor_solve_device(r_vals.ptr, r_eta.ptr) != 0:
                raise SystemExit("resident GN: factorization failed")
            return ctx_.ba_update_device(nc_, d_cw.ptr, d_s["cam_dxoff"].ptr, np_, d_pw.ptr, d_s["pt_dxoff"].ptr,
                                         r_eta.ptr, st_.n, apply=True)
        what = ("device linearization (CEdgeP2C3D, analytic) + assembly + Schur solve + ||dx|| + vertex update; "
                "state reset by two device copies per iteration (included)")
        return iterate, what, linearize

    # ---- third loop (one GPU): the whole (damped) Gauss-Newton iteration in HBM -- device linearization in the
    # reference's parameterization, assembly, solve, ||dx|| and the vertex update; the state is reset from a device copy
    # every iteration so that all iterations do the same work. Host traffic per iteration: 8 bytes.
    resident = None
    batch5 = None
    if world == 1 and "geometry" in prob:
        gn_resident, what, linearize = build_resident(ctx, st)
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
        resident = {"iters_per_s": gn_steps / dt_res, "ms_per_iter": 1e3 * dt_res / gn_steps, "dx_norm": dxn, "what": what}
        if linearize is not None:
            # the linearization kernel alone (hipEvent-free: wall clock around a synchronized burst)
            t0 = time.perf_counter()
            for _ in range(5):
                linearize()
            ctx.synchronize()
            lin_ms = 1e3 * (time.perf_counter() - t0) / 5
            resident["linearize_ms"] = lin_ms
            resident["linearize_gbs"] = (prob.v0.size * (160 + 16 + 8) + 0.0) / (lin_ms * 1e-3) * 1e-9
        # ---- SURVEY 8d metric 2: the batch of 5 iterations END TO END on a fresh context, analysis included -- what the
        # reference's Optimize() does once per batch (symbolic once, LinearSolver_UberBlock.h:317-320; 5 iterations,
        # src/slam_app/Main.cpp:706-707; yardstick scripts/tests/unit_tests.sh:50-56). Host arrays (graph, Jacobian inputs) in
        # host memory when the clock starts; the process-wide one-time set-up (code objects, stream self-tests) is warm.
        try:
            t0 = time.perf_counter()
            c5 = api.Context(local_rank, 0)
            t1 = time.perf_counter()
            st5 = c5.assemble_analyze(prob.dim, prob.v0, prob.v1, prob.d0, prob.d1, prob.rd, prob.unary_vertex)
            t2 = time.perf_counter()
            c5.analyze(st5, api.MODE_AUTO)
            t3 = time.perf_counter()
            it5, _, _ = build_resident(c5, st5)
            c5.synchronize()
            t4 = time.perf_counter()
            for i in range(5):
                dx5 = it5(reset=(i == 0))
            c5.synchronize()
            t5 = time.perf_counter()
            batch5 = {"total_ms": 1e3 * (t5 - t0), "context_ms": 1e3 * (t1 - t0), "assemble_analyze_ms": 1e3 * (t2 - t1),
                      "analyze_ms": 1e3 * (t3 - t2), "upload_state_ms": 1e3 * (t4 - t3), "five_iterations_ms": 1e3 * (t5 - t4),
                      "last_dx_norm": dx5,
                      "what": "fresh context: Lambda structure + assembly plan from the graph (spp_assemble_analyze), symbolic analysis "
                              "(spp_analyze: ordering / supernodes or the Schur plan), upload of the graph state, then 5 resident "
                              "iterations (linearize, assemble, solve, update), end to end"}
            c5.close()
        except Exception as e:  # noqa: BLE001
            batch5 = {"failed": repr(e)}

    # ---- N > 1, strong headline: the weak-scaling variant as an extra field (its own problem per rank, own ctx)
    weak_extra = None
    if world > 1 and not weak and args.workload == "venice871" and not args.graph:
        probw = synth.ba_problem(871, 530304, 2838740, 871 + rank, heavy_tail=True, name="venice871")
        jobw = Job(probw, (0, 1), 0)
        ksteps = max(3, args.steps // 2)
        dtw = jobw.timed(ksteps, 2)
        weak_extra = {"value": jobw.st.nnzb * world * ksteps / dtw, "unit": "block-nnz/s", "ms_per_step": 1e3 * dtw / ksteps,
                      "steps": ksteps, "what": "every rank brings its own Venice-sized landmark shard (530304 points, 2838740 "
                      "observations, seed 871 + rank) seen by the same 871 cameras: ONE bundle adjustment with %d x 530304 "
                      "landmarks, per-GPU work fixed; value = sum of the shards' block-nnz / time" % world}
        del jobw

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    x = x_solve
    total_nnzb = nnzb * world if weak else nnzb
    mode_name = {api.MODE_SCHUR: "schur+dense", api.MODE_SCHUR_SPARSE: "schur+sparse reduced system",
                 api.MODE_SPARSE: "sparse multifrontal"}[mode]
    parallelism = "single GPU"
    if world > 1 and schur:
        rep = phase["factor"] + phase["trisolve"]                      # replicated on every rank
        shard = phase["schur_inv"] + phase["schur_gemm"] + phase["schur_rhs"] + phase["backsubst"]  # this rank's share
        parallelism = ("%s: landmark shards x%d + one RCCL all-reduce of the packed reduced camera system (%.0f MB) on the ctx "
                       "stream; per solve %.2f ms replicated (dense factor + triangular solves) + %.2f ms sharded on this rank "
                       "(x%d = %.2f ms of single-GPU work): Amdahl bound %.2fx before the collective" % (
                           "weak (871 cameras, 530304 landmarks per GPU)" if weak else "strong (the one Venice problem)", world,
                           8e-6 * ctx.schur_packed_size(), rep, shard, world, shard * world,
                           (rep + shard * world) / max(rep, 1e-9)))
    elif world > 1:
        parallelism = "replicas only: a pose graph of this size does not shard (DESIGN.md 6); every rank solves the same problem"
    out = {
        "metric": METRIC, "value": total_nnzb * args.steps / dt, "unit": "block-nnz/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64",
        "data": "graph file" if args.graph else "synthetic",
        "config": {"workload": graph_note if graph_note else ("%s-shaped synthetic BA (%d cams, %d points, %d observations, LM-damped)" % (
            args.workload, prob.get("nc", 0), prob.get("npts", 0), prob.v0.size) if "nc" in prob else args.workload),
            "nnzb": int(nnzb), "n": int(st.n), "mode": mode_name,
            "n_reduced": int(ctx.info("N_REDUCED")), "schur_pairs": int(ctx.info("SCHUR_PAIRS")),
            "parallelism": parallelism},
        "gn_iters_per_s": gn_steps / dt_gn, "ms_per_gn_iter": 1e3 * dt_gn / gn_steps, "assemble_ms": assemble_ms,
        "phase_ms": {k: round(v, 4) for k, v in phase.items()},
        "phase_ms_note": "separate pass with SPP_FLAG_PROFILE on (hipEvents per phase and per trailing-update launch); ms_per_step is timed with profiling off", "analyze_s": round(analyze_s, 3),
        "generate_s": round(gen_s, 2), "solution_norm": float(np.linalg.norm(x)),
    }
    if world == 1:
        out["scaling_note"] = "one GPU: the whole problem (strong and weak coincide)"
    if schur:
        # what the run's own phase times predict for N GPUs of one node (strong scaling of THIS problem): the dense factor and
        # the triangular solves are replicated, the landmark phases divide, the exchange moves the packed reduced system
        # (direct: 1/N of it per point-to-point xGMI link each way, 153 GB/s per link -- MI355X guide; ring: 2 (N-1)/N of it
        # over one link). To be read against the driver's SCALE record.
        rep = phase["factor"] + phase["trisolve"]
        shard = (phase["schur_inv"] + phase["schur_gemm"] + phase["schur_rhs"] + phase["backsubst"]) * (world if world > 1 else 1)
        try:
            pk_bytes = 8.0 * ctx.schur_packed_size()
        except Exception:  # noqa: BLE001
            pk_bytes = 0.0
        link, lat_ms = 153e9, 0.03
        model = {"replicated_ms": rep, "sharded_ms_on_one_gpu": shard, "exchange_mb": pk_bytes * 1e-6, "per_link_gbs": 153.0,
                 "predicted_ms_direct": {}, "predicted_ms_ring": {}, "one_gpu_ms": rep + shard}
        for n_ in (2, 4, 8):
            model["predicted_ms_direct"][str(n_)] = rep + shard / n_ + 2e3 * (pk_bytes / n_) / link + 2 * lat_ms
            model["predicted_ms_ring"][str(n_)] = rep + shard / n_ + 2e3 * (n_ - 1) / n_ * pk_bytes / link + lat_ms
        out["scaling_model"] = model
    if weak_extra is not None:
        out["weak_scaling"] = weak_extra
    if resident is not None:
        out["gn_resident"] = resident
    if batch5 is not None:
        out["batch5_wall_ms"] = batch5

    # ---- HBM counter traffic of the dominant kernel: a rocprofv3 --pmc run of an EARLIER round (separate passes, the
    # guide's corrections), not of this run; the source file is named beside the number
    def pmc_traffic(key):
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            rec = json.load(open(pmc)).get(key)
            if not rec:
                return None, "no PMC record for this workload in profiles/pmc_traffic.json"
            return rec.get("bytes_per_launch"), "profiles/pmc_traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %s, not this run)" % (
                key, rec.get("source", "an earlier profiling run"))
        except Exception:
            return None, "no PMC record for this workload under profiles/"

    nobs, npair, nlm, nposes = ctx.info("N_OBS"), ctx.info("SCHUR_PAIRS"), ctx.info("N_LANDMARKS"), ctx.info("N_POSES")
    sch_ms = phase["schur_inv"] + phase["schur_gemm"] + phase["schur_rhs"] + phase["backsubst"]
    schur_hbm = None
    if schur and sch_ms > 0:
        # SURVEY 8d, "Schur": read 144 no + 72 np + 288 nnzb(A) + 8 n; write 288 nnzb(S) + 8 n  (algorithmic: every
        # input read once, every output written once; what the pair lists re-gather is NOT counted)
        nnzb_A = int(nnzb - nobs - nlm)
        sch_bytes = 144.0 * nobs + 72.0 * nlm + 288.0 * nnzb_A + 8.0 * st.n + 288.0 * ctx.info("S_NNZB") + 8.0 * st.n
        schur_hbm = {"achieved": sch_bytes / (sch_ms * 1e-3) * 1e-9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": sch_bytes / (sch_ms * 1e-3) * 1e-9 / PEAK_HBM_GBS, "algorithmic_bytes": sch_bytes, "ms": sch_ms,
                     "bytes_touched_by_pair_lists": 288.0 * npair,
                     "note": "SURVEY 8d bytes (inputs once, outputs once) over the four Schur phases; the S accumulation gathers "
                             "288 B per block product through its pair lists (bytes_touched_by_pair_lists), which the synthetic "
                             "co-visibility (observers drawn from a 209-camera window) does not let a cache absorb"}
    # ---- roofline of the dominant kernel
    # (a reduced system of a few tile rows -- Ladybug-49: 294 x 294 -- also goes through the streamed launch, but there the
    # landmark elimination is what is left to price: the HBM branch below)
    if mode == api.MODE_SCHUR and dom_n > 0 and dom_ms > 0 and ctx.info("N_REDUCED") >= 1024:
        achieved = dom_flops / (dom_ms * 1e-3) * 1e-12
        traffic, traffic_src = pmc_traffic(args.workload)
        n_red = float(ctx.info("N_REDUCED"))
        fac_flops = n_red ** 3 / 3.0
        mfma_meas = ctx.microbench_mfma_f64(4000)
        streamed = int(ctx.info("DENSE_STREAMED"))
        dom_name = ("spp::dense_tail_kernel (the streamed dense factor: one workgroup per 128x128 tile, rank-16 MFMA f64 16x16x4 updates as the "
                    "factorization's row tiles appear, diagonal tiles factored in LDS; %d of %d tile rows in this launch)" % (
                        streamed, (int(ctx.info("N_REDUCED")) + 127) // 128)) if streamed > 0 and dom_n == 1 else (
                    "spp::gemm_tn_mixed_kernel (MFMA f64 16x16x4 trailing update of the dense factor: 128x128 tiles, the tail of every launch cut into 64x64 quarters)")
        out["roofline"] = {"bound": "mfma", "kernel": dom_name,
                           "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                           "launches_per_solve": int(dom_n), "avg_launch_ms": dom_ms / dom_n,
                           "flops_per_launch": dom_flops / dom_n,
                           "factor_phase": {"flops": fac_flops, "ms": phase["factor"], "achieved": fac_flops / (phase["factor"] * 1e-3) * 1e-12,
                                            "frac": fac_flops / (phase["factor"] * 1e-3) * 1e-12 / PEAK_FP64_MFMA_TFLOPS,
                                            "frac_of_measured_mfma_loop": fac_flops / (phase["factor"] * 1e-3) * 1e-12 / mfma_meas,
                                            "what": "n_reduced^3 / 3 flops over the WHOLE dense factorization (serial diagonal-block chain included)"},
                           "whole_step": {"flops": fac_flops + 216.0 * npair, "ms": 1e3 * dt / args.steps,
                                          "frac": (fac_flops + 216.0 * npair) / (dt / args.steps) * 1e-12 / PEAK_FP64_MFMA_TFLOPS},
                           "standalone_update_tflops": (lambda m: (128.0 * m * (m + 1) + 256.0 * m) / ctx.microbench_update(m, 5) * 1e-9)(5120),
                           "measured_mfma_f64_peak": mfma_meas,
                           "measured_copy_gbs": ctx.microbench_copy(1 << 30, 10),
                           "measured_ctile_rw_gbs": ctx.microbench_ctile(8192, 5)}
        if schur_hbm:
            out["schur_hbm"] = schur_hbm
    elif mode == api.MODE_SCHUR and schur_hbm:
        # a reduced camera system too small for the MFMA trailing-update kernel (Ladybug-49: 294 x 294, factored inside one
        # workgroup): the landmark elimination is what is left to price, against HBM
        traffic, traffic_src = pmc_traffic(args.workload)
        out["roofline"] = dict(schur_hbm, bound="hbm", kernel="the four Schur phases (spp::cinv_kernel, obs_kernel, s_accum_kernel, rhs / "
                               "back-substitution kernels); the %d x %d reduced system is factored by single-workgroup kernels" % (
                                   ctx.info("N_REDUCED"), ctx.info("N_REDUCED")), traffic=traffic, traffic_source=traffic_src,
                               dense_factor={"flops": float(ctx.info("N_REDUCED")) ** 3 / 3.0, "ms": phase["factor"] + phase["trisolve"]})
    elif mode == api.MODE_SCHUR_SPARSE and schur_hbm:
        # BASELINE config 5 shape: the reduced camera system is sparse and small next to the landmark elimination
        traffic, traffic_src = pmc_traffic(args.workload)
        out["roofline"] = dict(schur_hbm, bound="hbm", kernel="the four Schur phases (spp::cinv_kernel, obs_kernel, s_accum_kernel, rhs / "
                               "back-substitution kernels): HBM-bound landmark elimination", traffic=traffic, traffic_source=traffic_src,
                               sparse_factor={"flops": int(ctx.info("FACTOR_FLOPS")), "ms": phase["factor"] + phase["trisolve"]})
    elif mode == api.MODE_SPARSE:
        # pose graphs: supernodal multifrontal factor + solves; bytes = 8 (nnz Lambda + nnz R) + 16 nnz R
        fs_ms = phase["factor"] + phase["trisolve"]
        sbytes, fflops = float(ctx.info("SOLVE_BYTES")), float(ctx.info("FACTOR_FLOPS"))
        if fs_ms > 0:
            traffic, traffic_src = pmc_traffic(args.workload)
            out["roofline"] = {"bound": "hbm", "kernel": "spp::front_dag_kernel / front_bwd_dag_kernel (supernodal multifrontal factor + backward "
                               "substitution, the whole assembly tree in ONE launch each: a workgroup per front -- a team of "
                               "workgroups per big front -- waits for its children's / its parent's flag; the forward "
                               "substitution rides through the factorization)",
                               "achieved": sbytes / (fs_ms * 1e-3) * 1e-9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": sbytes / (fs_ms * 1e-3) * 1e-9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes": sbytes, "factor_flops": fflops, "ms": fs_ms,
                               "mfma_frac": fflops / (max(phase["factor"], 1e-9) * 1e-3) * 1e-12 / PEAK_FP64_MFMA_TFLOPS,
                               "tree_levels": int(ctx.info("N_LEVELS")), "supernodes": int(ctx.info("N_SUPERNODES")),
                               "critical_path": {"levels": int(ctx.info("N_LEVELS")),
                                                 "factor_us_per_level": 1e3 * phase["factor"] / max(1, int(ctx.info("N_LEVELS"))),
                                                 "backward_us_per_level": 1e3 * phase["trisolve"] / max(1, int(ctx.info("N_LEVELS"))),
                                                 "what": "the dependency-driven launches are bound by the heaviest root-to-leaf chain of "
                                                         "fronts: time / tree levels = what one front of that chain costs (extend-add, "
                                                         "16-pivot panels at ~240 cycles per pivot, write-back, flag hand-off); "
                                                         "SPP_DAG_TRACE prints the chain front by front"},
                               "note": "LATENCY-bound, not bandwidth-bound: a few MB of factor whose critical path is the heaviest "
                                       "root-to-leaf chain of fronts over %d tree levels; neither roofline is approached at this "
                                       "size (SURVEY 7.3, 8d)" % int(ctx.info("N_LEVELS"))}
    # ---- the drop-in boundary's host-pointer entry (spp_factor_solve): Lambda in the ctx's page-locked staging buffer
    # (where the adapter flattens it) -> H2D -> solve -> D2H of the solution; never `value`
    if world == 1:
        try:
            lam_host = d_vals.download()
            eta_host = d_eta.download()
            t0 = time.perf_counter()
            stage = ctx.host_staging(st.nvals)
            # stands for the adapter's flatten (include/spp_adapter.h: per-block memcpy, a contiguous range of block
            # columns per host thread from 32 MB on): the same bytes by the same number of threads
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(16, os.cpu_count() or 1))
            cuts = [st.nvals * i // nthr for i in range(nthr + 1)]

            def _copy(i):
                stage[cuts[i]:cuts[i + 1]] = lam_host[cuts[i]:cuts[i + 1]]
            with ThreadPoolExecutor(nthr) as pool:
                list(pool.map(_copy, range(nthr)))
            flat_ms = 1e3 * (time.perf_counter() - t0)
            ctx.factor_solve(stage, eta_host)          # warm
            t0 = time.perf_counter()
            for _ in range(3):
                code, xd = ctx.factor_solve(stage, eta_host)
            drop_ms = 1e3 * (time.perf_counter() - t0) / 3
            out["dropin_ms"] = {"h2d_solve_d2h": drop_ms, "flatten_memcpy": flat_ms, "flatten_threads": nthr, "lambda_mb": 8e-6 * st.nvals,
                                "rel_diff_vs_resident": float(np.linalg.norm(xd - x) / np.linalg.norm(x)),
                                "what": "spp_factor_solve from the ctx's page-locked staging buffer (spp_host_staging): PCIe-inclusive, never `value`"}
        except Exception as e:  # noqa: BLE001
            out["dropin_ms"] = {"failed": repr(e)}
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
                # the REAL adapter at this size (part of the reference-side leg: oracle/_ref/dropin_driver is the reference's
                # CUberBlockMatrix + include/spp_adapter.h): Flatten_Values over 3.2 M blocks with up to 16 host threads
                drv = os.path.join(ROOT, "oracle", "_ref", "dropin_driver")
                if schur and os.path.exists(drv) and "nc" in prob and isinstance(out.get("dropin_ms"), dict):
                    import subprocess
                    track = max(2, int(round(prob.v0.size / max(1, prob.npts))))
                    r = subprocess.run([drv, "adapter", str(prob.nc), str(prob.npts), str(track)], capture_output=True, text=True, timeout=600)
                    f = dict(zip(r.stdout.split()[1::2], r.stdout.split()[2::2])) if r.returncode == 0 else {}
                    if f:
                        out["dropin_ms"]["adapter"] = {
                            "flatten_ms": float(f["flatten_ms"]), "factor_solve_ms": float(f["factor_solve_ms"]),
                            "threads": int(f["threads"]), "lambda_mb": float(f["lambda_mb"]), "blocks": int(f["blocks"]),
                            "rel_residual": float(f["rel_residual"]),
                            "what": "CLinearSolver_HIP::Solve_PosDef_Blocky on a CUberBlockMatrix of this shape (%s poses, %s landmarks, "
                                    "%d observers each) built through the reference's container: Flatten_Values + spp_factor_solve" % (
                                        f["cams"], f["points"], track)}
                out["cpu_baseline"]["headline_backend"] = (
                    "the reference's FASTEST path for this workload (%s); north_star's CHOLMOD path is the `cholmod` entry" % backend)
                # ---- north_star's CPU path: CLinearSolver_CholMod (src/slam/LinearSolver_CholMod.cpp:264-358: analyze + factorize
                # + solve on the whole Lambda, no Schur complement, analysis redone every call), same box, same run
                if args.cpu_cholmod != "off" and backend != "cholmod":
                    full = args.cpu_cholmod in ("full", "auto") or not schur or nnzb <= 200000
                    if full:
                        code, xc, sec = orc.RefSolver("cholmod", lam).solve(lam.vals, eta)
                        out["cpu_baseline"]["cholmod"] = {
                            "value": nnzb / sec, "unit": "block-nnz/s", "cores": 1, "seconds": sec,
                            "sample": "one full Lambda-solve of the same %s system by the reference's CLinearSolver_CholMod" % args.workload,
                            "rel_diff_gpu_vs_cholmod": float(np.linalg.norm(x - xc) / np.linalg.norm(xc))}
                    else:
                        # bounded sample: the same generator with a quarter of the landmarks (all 871 cameras), assembled by the HIP
                        # kernels in a second ctx; the whole Venice-shaped Lambda takes ~150 s per CHOLMOD solve
                        ps = synth.ba_problem(prob.nc, prob.npts // 4, prob.v0.size // 4, 871, heavy_tail=True, name="venice871_quarter")
                        js = Job(ps, (0, 1), 0)
                        lam_s, eta_s = js.st.with_vals(js.d_vals.download()), js.d_eta.download()
                        js.solve()
                        xs = js.d_rhs.download()
                        del js
                        code, xc, sec = orc.RefSolver("cholmod", lam_s).solve(lam_s.vals, eta_s)
                        out["cpu_baseline"]["cholmod"] = {
                            "value": lam_s.nnzb / sec, "unit": "block-nnz/s", "cores": 1, "seconds": sec,
                            "sample": "one Lambda-solve by the reference's CLinearSolver_CholMod of a QUARTER-size sample of the workload "
                                      "(same generator: 871 cameras, %d points, %d observations, %d upper blocks); the whole Venice-shaped "
                                      "Lambda takes 150-154 s per CHOLMOD solve (2.2e4 block-nnz/s, measured 2026-10-04 in the development "
                                      "container, 1 thread; --cpu-cholmod full repeats it here)" % (ps.npts, ps.v0.size, lam_s.nnzb),
                            "rel_diff_gpu_vs_cholmod": float(np.linalg.norm(xs - xc) / np.linalg.norm(xc))}
            else:
                out["cpu_baseline"] = {"value": None, "unit": "block-nnz/s", "cores": 1, "kind": "reference",
                                       "sample": "oracle/_ref/libspp_ref.so not present"}
        except Exception as e:  # the baseline must never take the measurement down
            out["cpu_baseline"] = {"value": None, "unit": "block-nnz/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
    elif not args.no_cpu_baseline and world == 1 and args.cpu_cholmod != "off":
        # BASELINE config 5 shape: the reference's Schur path is dense (60 000^2 does not fit it), its CHOLMOD path takes
        # minutes on the whole Lambda: a bounded sample of the SAME generator (1/8 of the cameras, points and observations,
        # the same ~5-camera co-visibility window), assembled by the HIP kernels in a second ctx
        try:
            from oracle import spp_oracle as orc
            if orc.have_ref() and "nc" in prob:
                f = 8
                ps = synth.ba_problem(prob.nc // f, prob.npts // f, prob.v0.size // f, 10000, heavy_tail=False,
                                      spread=0.0005 * f, name="%s_eighth" % args.workload)
                js = Job(ps, (0, 1), 0)
                lam_s, eta_s = js.st.with_vals(js.d_vals.download()), js.d_eta.download()
                js.solve()
                xs = js.d_rhs.download()
                t_gpu = js.timed(5, 1) / 5
                del js
                code, xc, sec = orc.RefSolver("cholmod", lam_s).solve(lam_s.vals, eta_s)
                out["cpu_baseline"] = {
                    "value": lam_s.nnzb / sec, "unit": "block-nnz/s", "cores": 1, "kind": "reference", "seconds": sec,
                    "sample": "one Lambda-solve by the reference's CLinearSolver_CholMod of a 1/8-size sample of the workload (same "
                              "generator and co-visibility window: %d cameras, %d points, %d observations, %d upper blocks); the "
                              "reference's Schur path is dense and does not fit the full 60 000-wide reduced system" % (
                                  ps.nc, ps.npts, ps.v0.size, lam_s.nnzb),
                    "gpu_ms_on_the_same_sample": 1e3 * t_gpu,
                    "rel_diff_gpu_vs_cholmod": float(np.linalg.norm(xs - xc) / np.linalg.norm(xc))}
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "block-nnz/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
    if world > 1 and job.S is not None:
        out["config"]["collective"] = " / ".join([job.xch.mode] + job.xch.notes)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
