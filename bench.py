#!/usr/bin/env python3
"""Benchmark of the flow-matching enhancement sampler hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the N=5 Euler sampler (prior sample ->
5 x [NCSN++ forward + fused Euler update]) on BASELINE.json config[1], a batch of 8 synthetic complex
spectrograms [8,1,256,256], fp32, synthetic (deterministic) weights of the released 65.6 M-parameter
architecture.  Inputs are resident in HBM before the timed region.  With N > 1 GPUs every rank runs its own
batch (per-utterance data parallel, weak scaling); the only collective is the final RCCL gather of the
enhanced spectrograms.  Rank 0 prints ONE JSON line.

Launch forms (both give one process per GPU over RCCL):
    python bench.py --gpus N ...                      bench.py spawns N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...    (the driver's form)
A run can never silently measure fewer GPUs than asked: N ranks need N visible devices, unless the test hook
FLOWSE_BENCH_SHARE_GPU=1 maps every rank to device 0 (then FLOWSE_BENCH_BACKEND=gloo, RCCL cannot share a device).

Metric: enhanced spectrogram-frames/sec at N=5 solver steps (frame = one STFT column of 256 bins).

`--workload vbdmd` runs BASELINE config[3] instead: 824 synthetic utterances of ragged length (T = 64 k, k in 2..10,
the set SURVEY 8(d) prescribes when the VoiceBank-DEMAND test set is absent) through flowmse_amd.parallel.
enhance_sharded -- equal-length batches dealt to the ranks by modelled time, N=5 Euler sampler, ONE final gather to rank 0 --
the multi-GPU form of the reference's loop over the test set (evaluate.py:97-136).  One "step" = one pass over the
whole set (total work fixed: "scaling": "strong"); the line carries every rank's frames and times, so load imbalance
and the serial gather tail are visible.  Runs at --gpus 1 as well.  `--workload vbdmd --plan --gpus N` prints the
partition only (per-rank utterances / batches / frames / batch fill / modelled time; host only, no GPU needed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FLOP_PER_FRAME_NFE = 2.080e9          # 532.57 GFLOP / 256 frames (T=256)
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 (= fp32 vector peak)
PEAK_16BIT_MATRIX_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (v_mfma_f32_32x32x16_*)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
FULL_CFG = dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), image_size=256)


def synth_state_dict(model):
    from flowmse_amd.util import synth
    return {n: torch.from_numpy(synth.synth_param(n, tuple(p.shape))) for n, p in model.named_parameters()}


def cpu_baseline(sd, nsolver, frames, reps, budget_s=60.0, oracle_utt=None):
    """Oracle (CPU restatement of the reference path, 'port') on the host cores: a bounded sample of the same
    workload -- ONE utterance [1,1,256,frames], ONE Euler step (1 NFE) -- scaled to N steps.  The torch thread
    count is chosen by a short sweep at 64 frames (more threads than physical cores can be much slower)."""
    from flowmse_amd.util import synth
    from oracle import ncsnpp_oracle as O
    from oracle import sampler_oracle as S
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:                                     # cgroup v2 CPU quota of the container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except Exception:
        pass
    cfg = O.make_cfg()
    t_start = time.perf_counter()

    def nfe_time(T, n):
        y = torch.from_numpy(synth.synth_spectrogram(0, 1, 256, T))
        z = torch.from_numpy(synth.synth_noise(0, 1, 256, T))
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            S.euler_sample_net(sd, cfg, y, z, N=1)
            ts.append(time.perf_counter() - t0)
        return ts

    sweep = {}
    best_n, best_t = None, None
    for n in [c for c in (8, 16, 32, 64, 96, 128, 192, 256) if c <= logical] or [logical]:
        torch.set_num_threads(n)
        t = min(nfe_time(64, 2))
        sweep[n] = round(t, 3)
        if best_t is None or t < best_t:
            best_n, best_t = n, t
        elif t > 2.0 * best_t or time.perf_counter() - t_start > budget_s / 2:
            break
    torch.set_num_threads(best_n)
    times = nfe_time(frames, reps + 1)[1:]
    t_nfe = sorted(times)[len(times) // 2]
    x_oracle = None
    if oracle_utt is not None:          # the checker's own N-step answer for ONE utterance of the timed batch (parity echo)
        y = torch.from_numpy(synth.synth_spectrogram(oracle_utt, 1, 256, frames))
        z = torch.from_numpy(synth.synth_noise(oracle_utt, 1, 256, frames))
        x_oracle = S.euler_sample_net(sd, cfg, y, z, N=nsolver)[0]
    return x_oracle, {"value": frames / (nsolver * t_nfe), "unit": "frames/s", "cores": best_n, "kind": "port",
            "host_logical_cpus": logical, "cgroup_cpu_quota": quota,
            "thread_sweep_s_per_nfe_at_64_frames": sweep,
            "sample": f"oracle (torch-CPU fp32 restatement of the reference path) on 1 utterance [1,1,256,{frames}], "
                      f"1 Euler step = 1 NFE, median of {reps} after 1 warm-up = {t_nfe:.3f} s/NFE with "
                      f"{best_n} torch threads (best of the sweep), scaled to N={nsolver} steps "
                      f"(the path is linear in batch and steps)"}


# BASELINE config[3] stand-in for the VoiceBank-DEMAND test set (824 utterances, public size of that set; the data
# itself is not available offline): padded frame counts T = 64 k, k = 2..10, skewed towards short
# utterances like the real set (most files 2-4 s), true lengths uniformly inside the last 64-frame block.
VBDMD_UTTS = 824
VBDMD_K_WEIGHTS = {2: 0.16, 3: 0.22, 4: 0.20, 5: 0.14, 6: 0.10, 7: 0.07, 8: 0.05, 9: 0.035, 10: 0.025}


def vbdmd_lengths(n):
    """Deterministic true frame counts of the n synthetic utterances (hash-based, no RNG library)."""
    from flowmse_amd.util import synth
    u = synth.uniform01(2024, 3, n)
    v = synth.uniform01(2024, 4, n)
    ks, acc, cdf = sorted(VBDMD_K_WEIGHTS), 0.0, []
    for k in ks:
        acc += VBDMD_K_WEIGHTS[k]
        cdf.append(acc)
    out = []
    for a, b in zip(u, v):
        k = next((kk for kk, c in zip(ks, cdf) if a * acc < c), ks[-1])
        out.append(64 * (k - 1) + 1 + int(b * 64))            # in (64 (k-1), 64 k]
    return out


def vbdmd_plan(n, world, max_batch):
    """Host-only: the partition of the config[3] stand-in set over `world` ranks and its summary."""
    from flowmse_amd.parallel import plan_shards, plan_summary
    true_len = vbdmd_lengths(n)
    padded = [((t + 63) // 64) * 64 for t in true_len]
    plan = plan_shards(padded, world, max_batch)
    return true_len, padded, plan, plan_summary(plan, padded, max_batch)


def run_vbdmd(args, model, dev, world, rank, backend, share, share_of=None):
    """config[3]: the whole ragged set through enhance_sharded, timed end to end (batches -> deal -> sampler -> gather).

    share_of = (r, W) on ONE GPU: run exactly rank r's share of the W-rank partition of the whole set (what one GPU of
    a W-GPU job executes), not a smaller set."""
    from flowmse_amd.parallel import enhance_sharded
    from flowmse_amd.sampling import get_white_box_solver
    from flowmse_amd.util import synth
    n, F, NS = args.utts, 256, args.nsolver
    if share_of is None:
        true_len, padded, plan, psum = vbdmd_plan(n, world, args.batch)
    else:
        assert world == 1
        true_len, padded, plan_w, psum_w = vbdmd_plan(n, share_of[1], args.batch)
        plan = [plan_w[share_of[0]]]
        from flowmse_amd.parallel import plan_summary
        psum = plan_summary(plan, padded, args.batch)
    mine = set(i for _, ids in plan[rank] for i in ids)
    # every rank holds the lengths of all utterances and the data of its own shard, resident in HBM before timing
    specs, noise = [], {}
    for i in range(n):
        if i in mine:
            specs.append(torch.from_numpy(synth.synth_spectrogram(i, 1, F, true_len[i]))[0, 0].to(dev))
            noise[i] = torch.from_numpy(synth.synth_noise(i, 1, F, padded[i])).to(dev)
        else:
            specs.append(torch.empty(F, true_len[i], dtype=torch.complex64, device="meta"))
    ws_bytes = model.dnn.reserve(min(args.batch, n), F, max(padded))

    def sample_fn(Y, ids):
        Z = torch.cat([noise[i] for i in ids])
        return get_white_box_solver(args.solver, model.ode, model, Y=Y, Y_prior=Y, T_rev=1.0, t_eps=0.03, N=NS, z=Z)()[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass(stats):
        return enhance_sharded(sample_fn, specs, max_batch=args.batch, stats=stats, plan=plan)

    for _ in range(max(args.warmup, 2)):          # every (batch, T) shape: eager pass, then hipGraph capture
        one_pass({})
    g0 = model.dnn.graph_launches()
    barrier()
    t0 = time.perf_counter()
    st = {}
    for _ in range(args.steps):
        st = {}
        out = one_pass(st)
    barrier()
    elapsed = time.perf_counter() - t0
    graph_launches = model.dnn.graph_launches() - g0
    mine_row = [elapsed, st["sample_s"], st["gather_s"], float(st["frames"]), float(st["utterances"]), float(st["batches"]),
                float(st["batch_fill"])]
    rows = [mine_row]
    if world > 1:
        tt = torch.zeros(world, len(mine_row), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tt[rank] = torch.tensor(mine_row, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        rows = tt.tolist()
        elapsed = max(r[0] for r in rows)
    if rank != 0:
        return None
    got = sorted(mine) if share_of is not None else range(n)
    assert len(out) == n and all(out[i] is not None and out[i].shape == (F, true_len[i]) for i in got), "gather lost data"
    checked = not os.environ.get("FLOWSE_BENCH_NO_CHECK")
    if checked:
        assert all(bool(torch.isfinite(torch.view_as_real(out[i])).all()) for i in got), "non-finite output"
    frames = sum(padded[i] for i in got)
    n_run = len(got)
    nfe_per_step = (NS - 1) * {"euler": 1, "heun": 2, "rk4": 4}[args.solver] + 1
    value = args.steps * frames / elapsed
    hist = {}
    for i in got:
        hist[padded[i]] = hist.get(padded[i], 0) + 1
    rank_frames = [r[3] for r in rows]
    return {
        "metric": f"enhanced spectrogram-frames/sec at N={NS} solver steps",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
        "config": {"workload": f"BASELINE config[3]: {n} synthetic utterances standing in for the VoiceBank-DEMAND test set "
                               f"(not available offline), ragged [1,1,{F},T] with padded T = 64k, k in 2..10, "
                               + (f"rank {share_of[0]}'s share ({n_run} utterances) of the {share_of[1]}-rank partition, "
                                  "run on one GPU" if share_of is not None else
                                  f"cut into equal-length batches of <= {args.batch} and dealt to {world} GPU(s) by "
                                  "modelled time (flowmse_amd.parallel.plan_shards)") +
                               f", N={NS} {args.solver} steps ({nfe_per_step} NFE), NCSN++ (65.6M params, synthetic weights), "
                               f"precision mode {args.precision}; one step = one pass over the set incl. the final "
                               "gather to rank 0",
                   "utterances": n_run, "frames_padded_total": frames, "frames_true_total": sum(true_len[i] for i in got),
                   "padded_length_histogram": {str(k): hist[k] for k in sorted(hist)},
                   "max_batch": args.batch, "solver_steps": NS,
                   "parallelism": f"dp{world} (per-utterance, final gather to rank 0 only)",
                   "collective_backend": (backend if world > 1 else None), "dist": dist_info(world, backend),
                   "ranks_share_one_device": share if world > 1 else False,
                   "launch_mode": launch_mode(graph_launches), "hip_graph_launches_rank0": graph_launches,
                   "finite_output_checked": checked,
                   "workspace_GB": round(ws_bytes / 2 ** 30, 2)},
        "frames_value_counts": "padded frames (what the kernels process); true (unpadded) frames/s = value * "
                               f"{sum(true_len[i] for i in got) / frames:.4f}",
        "plan": {k: psum[k] for k in ("frame_imbalance_max_over_mean", "model_time_imbalance_max_over_mean", "batch_fill",
                                      "promotion_padding_frames")},
        "per_rank": {"frames": rank_frames, "utterances": [r[4] for r in rows], "batches": [r[5] for r in rows],
                     "batch_fill": [round(r[6], 4) for r in rows],
                     "ms_per_step": [round(1e3 * r[0] / args.steps, 3) for r in rows],
                     "last_pass_sampler_ms": [round(1e3 * r[1], 3) for r in rows],
                     "last_pass_gather_ms": [round(1e3 * r[2], 3) for r in rows],
                     "frame_imbalance_max_over_mean": max(rank_frames) / (sum(rank_frames) / len(rank_frames))},
        "achieved_TFLOPs_whole_path": value * nfe_per_step * FLOP_PER_FRAME_NFE / 1e12,
    }


def n1_full_set_reference():
    """The committed 1-GPU full-set figure of the config[3] workload (profiles/rNN_vbdmd_1gpu_line.json, written by
    `python bench.py --workload vbdmd` on one MI355X), for `vs_N1_full_set` in a multi-GPU line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_vbdmd_1gpu_line.json")))
    if not files:
        return None
    j = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    return {"value": j["value"], "ms_per_step": j["ms_per_step"], "utterances": j["config"]["utterances"],
            "source": f"profiles/{os.path.basename(files[-1])} (the builder's 1-GPU run of the same workload, not measured "
                      "in this run)"}


def rank0_tail_probe(dev, world_of, max_batch, n=None):
    """Rank 0's device -> host tail of the config[3] gather at `world_of` ranks, measured on ONE GPU: the received slabs
    of every length bucket are faked on the device (same shapes and row tables as the real exchange) and drained through
    the product's own _HostDrain -- first call (allocates + page-locks the staging buffer), second call (reuses it) --
    next to the round-4 form (fresh pin_memory() per call + a .clone() per row) for the before/after."""
    from flowmse_amd import parallel as P
    n = n or VBDMD_UTTS
    true_len, padded, plan, _ = vbdmd_plan(n, world_of, max_batch)
    F = 256
    buckets = {}
    for r in range(world_of):
        for T, ids in plan[r]:
            for i in ids:
                buckets.setdefault(padded[i], [[] for _ in range(world_of)])[r].append((i, true_len[i]))
    slabs = {}
    for Tp, per_rank in buckets.items():
        kmax = max(len(v) for v in per_rank)
        slabs[Tp] = [torch.zeros((kmax, F, Tp, 2), dtype=torch.float32, device=dev) for _ in range(world_of)]
    total = sum(F * t * 2 for t in true_len)
    torch.cuda.synchronize()

    def product():
        t0 = time.perf_counter()
        out = [None] * n
        d = P._HostDrain(F, total, dev)
        for Tp in sorted(buckets, reverse=True):
            d.add([(slabs[Tp][r][row], i, t) for r in range(world_of) for row, (i, t) in enumerate(buckets[Tp][r])])
        d.finish(out)
        return 1e3 * (time.perf_counter() - t0), out

    def round4():
        t0 = time.perf_counter()
        stage = torch.empty(total, dtype=torch.float32).pin_memory()
        off, views, out = 0, [], [None] * n
        for Tp in sorted(buckets, reverse=True):
            for r in range(world_of):
                for row, (i, t) in enumerate(buckets[Tp][r]):
                    v = stage[off:off + F * t * 2].view(F, t, 2)
                    v.copy_(slabs[Tp][r][row, :, :t], non_blocking=True)
                    views.append((i, v))
                    off += F * t * 2
        torch.cuda.current_stream().synchronize()
        for i, v in views:
            out[i] = torch.view_as_complex(v.clone())
        return 1e3 * (time.perf_counter() - t0)

    first, out = product()
    del out
    second, out = product()
    del out
    third, out = product()
    ok = all(o is not None and o.shape == (F, true_len[i]) for i, o in enumerate(out))
    del out
    before = [round4(), round4()]
    return {"world_of": world_of, "utterances": n, "payload_MB": round(total * 4 / 1e6, 1),
            "product_ms": {"first_call_allocates_and_pins": round(first, 2), "second_call": round(second, 2),
                           "third_call": round(third, 2)},
            "round4_form_ms": [round(b, 2) for b in before], "rows_ok": ok,
            "staging_pool": dict(zip(("buffers", "bytes"), P.stage_pool_stats())),
            "note": "device -> host copies + host bookkeeping of rank 0 after the last gather, fake received slabs of the "
                    f"{world_of}-rank config[3] partition resident on one GPU; the RCCL gathers themselves are not in it"}


CONFIG4 = dict(batch=32, frames=1024, nsolver=25, solver="rk4", precision="fp16")


def run_config4(model, dev, min_free_gb=40.0):
    """BASELINE config[4] inside the default line: [32,1,256,1024], fixed-step RK4 N = 25 (97 NFE), fp16 storage with
    fp32 GroupNorm statistics / accumulation.  One warm-up pass -- bracketed with the library's per-launch HIP events
    around the 16-bit 3x3 conv, which gives this workload's own roofline block -- then ONE timed pass (~8 s each)."""
    from flowmse_amd.sampling import get_white_box_solver
    from flowmse_amd.util import synth
    c = CONFIG4
    B, F, T, NS = c["batch"], 256, c["frames"], c["nsolver"]
    free_gb = torch.cuda.mem_get_info(dev)[0] / 2 ** 30
    if free_gb < min_free_gb:
        return {"skipped": f"{free_gb:.1f} GB of device memory free, {min_free_gb:.0f} GB wanted"}
    prev = model.dnn.precision
    model.dnn.set_precision(c["precision"])
    try:
        ws_bytes = model.dnn.reserve(B, F, T)
        Y = torch.cat([torch.from_numpy(synth.synth_spectrogram(i, 1, F, T)) for i in range(B)]).to(dev)
        Z = torch.cat([torch.from_numpy(synth.synth_noise(i, 1, F, T)) for i in range(B)]).to(dev)

        def one_pass():
            return get_white_box_solver(c["solver"], model.ode, model, Y=Y, Y_prior=Y, T_rev=1.0, t_eps=0.03, N=NS, z=Z)()[0]

        model.dnn.profile_begin(0)
        x0 = one_pass()
        torch.cuda.synchronize()
        prof = model.dnn.profile_end()
        t0 = time.perf_counter()
        x1 = one_pass()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        finite = bool(torch.isfinite(torch.view_as_real(x1)).all())
        repeat_identical = bool(torch.equal(torch.view_as_real(x0), torch.view_as_real(x1)))
        assert finite, "config[4]: non-finite output"
    finally:
        model.dnn.set_precision(prev)
    nfe = (NS - 1) * 4 + 1
    value = B * T / dt
    res = {"value": value, "unit": "frames/s", "ms_per_step": 1e3 * dt, "steps": 1, "warmup": 1, "nfe_per_step": nfe,
           "dtype": DTYPE_NAMES[c["precision"]],
           "config": {"workload": f"BASELINE config[4]: batch={B} synthetic complex spectrograms [{B},1,{F},{T}], N={NS} "
                                  f"fixed-step rk4 steps ({nfe} NFE; the last step is the reference's Euler update onto t=0), "
                                  "NCSN++ (65.6M params, synthetic weights), fp16 storage + fp16 matrix-core operands, fp32 "
                                  "GroupNorm statistics / accumulation, one MI355X",
                      "workspace_GB": round(ws_bytes / 2 ** 30, 2)},
           "achieved_TFLOPs_whole_path": value * nfe * FLOP_PER_FRAME_NFE / 1e12,
           "finite_output_checked": finite, "second_pass_bit_identical": repeat_identical,
           "parity_note": "fixed-step rk4 has no reference counterpart (SURVEY 8 a19: the reference's RK slot is scipy RK45, "
                          "sampling/__init__.py:64-114): the tableau is pinned against the oracle-VF composition at small size "
                          "(tests/test_gpu_model.py); at this full size the run is property-checked only (finite, "
                          "bit-identical repeat)"}
    dm = prof.get("dominant_conv3x3")
    if dm and dm["ms"] > 0:
        tf = dm["issued"] / (dm["ms"] * 1e-3) / 1e12
        gbs = dm["bytes"] / (dm["ms"] * 1e-3) / 1e9
        res["roofline"] = {"kernel": "flowse::conv3x3_pc16_kernel (persistent producer/consumer LDS-halo 3x3 conv, 16-bit MFMA "
                                     "operands, fused GroupNorm+SiLU input, folded 1x1 shortcuts)",
                           "bound": "hbm" if gbs / HBM_PEAK_GBS > tf / PEAK_16BIT_MATRIX_TFLOPS else "mfma",
                           "achieved": tf, "peak": PEAK_16BIT_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_16BIT_MATRIX_TFLOPS,
                           "achieved_GBs_algorithmic": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                           "launches": dm["launches"], "avg_launch_ms": dm["ms"] / dm["launches"],
                           "flops_per_launch_avg": dm["flops"] / dm["launches"],
                           "algorithmic_bytes_per_launch_avg": dm["bytes"] / dm["launches"],
                           "measured_in": "the warm-up pass (HIP events on the launch stream around each launch of the kernel)"}
        tot = prof.get("_all_launches")
        if tot:
            res["launches_per_nfe"] = tot["launches"] / nfe
    return res


def launch_mode(graph_launches):
    return ("hipGraph replay (FLOWSE_GRAPH=1): one graph launch per network evaluation" if graph_launches > 0 else
            "eager launches of the per-shape launch list (the library's default)")


def dist_info(world, backend):
    """Backend facts echoed into the line when world > 1 (what an 8-GPU run would otherwise have to be re-run to learn)."""
    if world <= 1:
        return None
    info = {"backend": backend, "visible_devices": torch.cuda.device_count(),
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
            "NCCL_DEBUG": os.environ.get("NCCL_DEBUG")}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                                     # never let a version probe fail the run
            info["rccl_version"] = f"unavailable ({type(e).__name__})"
    return info


DTYPE_NAMES = {"fp32": "f32",
               "bf16x3": "f32 (3x3 convs as split-bf16 x3 MFMA, fp32 accumulate)",
               "bf16": "bf16 activation storage + bf16 matrix-core operands (fp32 accumulate, GroupNorm statistics, "
                       "softmax state and 4-channel tensors)",
               "fp16": "fp16 activation storage + fp16 matrix-core operands (fp32 accumulate, GroupNorm statistics, "
                       "softmax state and 4-channel tensors)"}


def spawn_ranks(n):
    """Re-execute this script as n ranks on this node (one per GPU) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--nsolver", type=int, default=5)
    ap.add_argument("--solver", default="euler", choices=["euler", "heun", "rk4"],
                    help="ODE solver plugin (BASELINE config 5: --solver rk4 --nsolver 25 --batch 32 --frames 1024 "
                         "--precision fp16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16", "fp16"],
                    help="matrix-core operand mode of the large 3x3 convs (default: exact fp32)")
    ap.add_argument("--no-alt", action="store_true", help="skip the bf16x3 / bf16 operand-mode legs")
    ap.add_argument("--profile-all", action="store_true", help="per-op timing table to stderr (extra untimed pass)")
    ap.add_argument("--workload", default="config1", choices=["config1", "vbdmd"],
                    help="config1: BASELINE config[1] (one [batch,1,256,frames] batch per GPU, weak scaling; the default); "
                         "vbdmd: BASELINE config[3] (ragged utterance set sharded over the GPUs, strong scaling)")
    ap.add_argument("--utts", type=int, default=VBDMD_UTTS, help="--workload vbdmd: number of utterances")
    ap.add_argument("--no-strong", action="store_true",
                    help="world > 1: skip the strong-scaling config[3] pass that is otherwise run after the timed region")
    ap.add_argument("--plan", action="store_true",
                    help="--workload vbdmd: print the partition over --gpus ranks (host only, no GPU needed) and exit")
    args = ap.parse_args()

    if args.plan:
        assert args.workload == "vbdmd", "--plan describes the --workload vbdmd partition"
        _, padded, plan, psum = vbdmd_plan(args.utts, args.gpus, args.batch)
        psum["workload"] = (f"BASELINE config[3] stand-in: {args.utts} utterances, {sum(padded)} padded frames, "
                            f"batches of <= {args.batch}, {args.gpus} rank(s)")
        psum["batches_per_rank"] = [[(T, len(ids)) for T, ids in b] for b in plan]
        print(json.dumps(psum), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:         # plain `python bench.py --gpus N`: spawn the ranks
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: refusing to measure a different GPU count"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    share = bool(os.environ.get("FLOWSE_BENCH_SHARE_GPU"))
    if share:                                                    # test hook: every rank on device 0
        local_rank = 0
    else:
        assert torch.cuda.device_count() >= world, (
            f"--gpus {world} needs {world} visible devices, found {torch.cuda.device_count()} "
            "(set FLOWSE_BENCH_SHARE_GPU=1 FLOWSE_BENCH_BACKEND=gloo to run the ranks on one device for testing)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("FLOWSE_BENCH_BACKEND", "nccl")     # "gloo": CI on a box where ranks share one GPU
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # the HIP library normally travels with the repo (built by __graft_entry__.build()); build it if it is missing
    from flowmse_amd import build as fb
    if not os.path.exists(fb.LIB):
        if local_rank == 0:
            fb.build(verbose=False)
        else:
            t_wait = time.time()
            while not (os.path.exists(fb.LIB) and os.path.exists(fb.LIB + ".stamp")):
                assert time.time() - t_wait < 600, "timed out waiting for rank 0 to build libflowse_hip.so"
                time.sleep(1.0)
    from flowmse_amd.model import VFModel
    from flowmse_amd.sampling import get_white_box_solver
    from flowmse_amd.util import synth

    model = VFModel(backbone="ncsnpp", ode="flowmatching", **FULL_CFG)
    sd = synth_state_dict(model.dnn)
    model.dnn.load_state_dict(sd)
    model = model.to(dev).eval()
    model.dnn.set_precision(args.precision)

    if args.workload == "vbdmd":
        out = run_vbdmd(args, model, dev, world, rank, backend, share)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    B, F, T, NS = args.batch, 256, args.frames, args.nsolver
    # this rank's batch: utterance indices rank*B .. rank*B+B-1 (seeds 1234+i / 4321+i)
    Y = torch.cat([torch.from_numpy(synth.synth_spectrogram(rank * B + i, 1, F, T)) for i in range(B)]).to(dev)
    Z = torch.cat([torch.from_numpy(synth.synth_noise(rank * B + i, 1, F, T)) for i in range(B)]).to(dev)
    ws_bytes = model.dnn.reserve(B, F, T)
    graphs_on = bool(os.environ.get("FLOWSE_GRAPH"))             # opt-in (slower)

    def step():
        sampler = get_white_box_solver(args.solver, model.ode, model, Y=Y, Y_prior=Y, T_rev=1.0, t_eps=0.03, N=NS, z=Z)
        x, n = sampler()
        return x

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            xx = step()
        if world > 1:                              # the path's only exchange: final gather of enhanced specs
            payload = torch.view_as_real(xx).contiguous()
            if backend != "nccl":
                payload = payload.cpu()
            gathered = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
            dist.gather(payload, gathered, dst=0)
        barrier()
        return xx, time.perf_counter() - t0

    for _ in range(max(args.warmup, 2 if graphs_on else 0)):   # a shape's hipGraph is captured on its second pass
        x = step()
    g0 = model.dnn.graph_launches()
    # (1) THE timed region: the product path as shipped -- plain (eager) launches of the shape's launch list by
    # default, one hipGraph replay per network evaluation only under FLOWSE_GRAPH=1; `launch_mode` in the line says
    # which one ran, from the library's own replay counter
    x, elapsed = timed_region()
    graph_launches = model.dnn.graph_launches() - g0          # counted by the library, not inferred from the env
    # (2) the same K steps again with the library's per-launch HIP events (recorded on the launch stream) around the
    # dominant kernel (plain launches of the identical launch list).  Its wall time is reported next to the timed
    # region's (roofline.profiled_region_*).
    model.dnn.profile_begin(0)
    _, elapsed_prof = timed_region()
    prof = model.dnn.profile_end()
    per_rank_ms = [1e3 * elapsed / args.steps]
    if world > 1:                                  # MAX over ranks is the job's time; keep every rank's own clock too
        tt = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        tt[rank] = elapsed
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank_ms = [1e3 * float(v) / args.steps for v in tt.tolist()]
        elapsed = float(tt.max().item())
    checked = not os.environ.get("FLOWSE_BENCH_NO_CHECK")      # measurement hook (what-if builds); echoed into the line
    assert not checked or torch.isfinite(torch.view_as_real(x)).all(), "non-finite output"

    frames_total = world * args.steps * B * T
    value = frames_total / elapsed
    # higher-order solvers take the reference's Euler update on the last step (it lands on t = 0): 1 NFE there
    nfe_per_step = (NS - 1) * {"euler": 1, "heun": 2, "rk4": 4}[args.solver] + 1
    out = {
        "metric": f"enhanced spectrogram-frames/sec at N={NS} solver steps",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
        "config": {"workload": f"BASELINE config[1]: batch={B} synthetic complex spectrograms [{B},1,{F},{T}] per GPU, "
                               f"N={NS} {args.solver} steps ({nfe_per_step} NFE), NCSN++ (65.6M params, synthetic "
                               f"weights), precision mode {args.precision}",
                   "global_batch": world * B, "frames": T, "solver_steps": NS,
                   "parallelism": f"dp{world} (per-utterance, final RCCL gather only)",
                   "collective_backend": (backend if world > 1 else None), "dist": dist_info(world, backend),
                   "ranks_share_one_device": share if world > 1 else False,
                   "launch_mode": launch_mode(graph_launches), "hip_graph_launches": graph_launches,
                   "finite_output_checked": checked,
                   "workspace_GB": round(ws_bytes / 2 ** 30, 2)},
        "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
        "achieved_TFLOPs_whole_path": value * nfe_per_step * FLOP_PER_FRAME_NFE / world / 1e12,
    }
    vb_strong = None
    if world > 1 and args.solver == "euler" and not args.no_strong:
        # The weak-scaling number above is ~N x by construction; the workload that shows what N GPUs buy is BASELINE
        # config[3]: the whole ragged set (fixed total work) dealt to the ranks, one final RCCL gather to rank 0.
        a3 = argparse.Namespace(**vars(args))
        a3.steps, a3.warmup = 1, 2
        vb_strong = run_vbdmd(a3, model, dev, world, rank, backend, share)        # every rank takes part (collectives)
    if rank == 0:
        dom = prof.get("dominant_conv3x3")
        if dom and dom["ms"] > 0:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            traffic, traffic_src = None, None
            import glob
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))
            if tfiles and (B, T, NS, args.precision) == (8, 256, 5, "fp32"):   # PMC passes were taken on this workload
                tj = json.load(open(tfiles[-1]))
                k = tj.get("dominant_kernel")
                if tj.get("dominant_set"):                  # launch-weighted over every instantiation bracketed here
                    traffic = tj["dominant_set"]["hbm_bytes_per_launch"]
                elif k:
                    traffic = tj["kernels"][k]["hbm_bytes_per_launch"]
                if traffic is not None:
                    traffic_src = (f"profiles/{os.path.basename(tfiles[-1])} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                   "own passes, 2*FETCH_SIZE + WRITE_SIZE)")
            form = None
            if args.precision == "fp32" and not os.environ.get("FLOWSE_NO_WINOGRAD"):
                form = "F(4,3)" if os.environ.get("FLOWSE_W2D", "1")[:1] == "0" else "F(4,3)xF(2,3)"
            kname = ("flowse::conv3x3_w2d_kernel<2> (fp32 two-dimensional Winograd F(4,3) x F(2,3) implicit-GEMM 3x3 conv, 16x16 "
                     "pixel x 64 channel block of 8 waves, LDS halo, fused GroupNorm+SiLU input, weights streamed from L2 in MFMA "
                     "fragment order; 32-channel blocks on launches of 128-255 such blocks), flowse::conv3x3_f43_kernel<2, false, *> (1-D F(4,3)) on "
                     "the smaller ones" if form == "F(4,3)xF(2,3)" else
                     "flowse::conv3x3_f43_kernel<2, false, 2>" +
                     f" (fp32 {form}-Winograd implicit-GEMM 3x3 conv, 8x16 pixel x 128 channel block, LDS halo, fused "
                     "GroupNorm+SiLU input, weights streamed from L2 in MFMA fragment order)" if form else
                     "flowse::conv3x3_halo_kernel<2,2,2,2,2> (fp32 implicit-GEMM 3x3 conv, 128x128 tile, LDS halo, "
                     "fused GroupNorm+SiLU input)" if args.precision == "fp32" else
                     "flowse::conv3x3_halo_bf16_kernel (fp32 storage, split-bf16 x3 operands, LDS halo)" if args.precision == "bf16x3" else
                     "flowse::conv3x3_pc16_kernel (+ conv3x3_halo_bf16_kernel on small launches): persistent producer/consumer "
                     "LDS-halo 3x3 conv, 16-bit MFMA operands, fused GroupNorm+SiLU input")
            # FLOPs the matrix cores execute for the bracketed launches, counted per launch by the library (1/3 of the direct
            # convolution's for the 2-D Winograd kernel, 1/2 for 1-D F(4,3), 3x for the split-bf16 mode)
            issued = dom["issued"] / (dom["ms"] * 1e-3) / 1e12
            issue = issued / ach
            peak = PEAK_FP32_MATRIX_TFLOPS if args.precision == "fp32" else PEAK_16BIT_MATRIX_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": kname,
                               "achieved": issued, "peak": peak, "unit": "TFLOP/s",
                               "frac": issued / peak,
                               "achieved_definition": "FLOPs the matrix cores execute per launch / launch time"
                                                      + (f": the {form} Winograd kernels issue {issue:.3g} of the algorithmic "
                                                         "direct-convolution FLOPs of these launches (1/3 per 2-D launch, 1/2 per "
                                                         "1-D launch; SURVEY 8d), which are reported as achieved_algorithmic"
                                                         if form else
                                                         " (= the algorithmic direct-convolution FLOPs of SURVEY 8d)"),
                               "achieved_algorithmic": ach,
                               "whole_path_algorithmic_TFLOPs": value * nfe_per_step * FLOP_PER_FRAME_NFE / world / 1e12,
                               "traffic": traffic, "traffic_source": traffic_src,
                               "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                               "flops_per_launch_avg": dom["flops"] / dom["launches"],
                               "algorithmic_bytes_per_launch_avg": dom["bytes"] / dom["launches"],
                               "time_share_of_step": dom["ms"] * 1e-3 / elapsed_prof,
                               "profiled_region_ms_per_step": 1e3 * elapsed_prof / args.steps,
                               "profiled_region_note": "HIP events bracket each launch of the kernel in a second "
                                                       "K-step region (plain launches of the identical launch list); "
                                                       "`value` comes from the first region (" +
                                                       ("hipGraph replay" if graph_launches > 0 else "eager launches too") + ")"}
            tot = prof.get("_all_launches")
            if tot:
                out["launches_per_nfe"] = tot["launches"] / (args.steps * nfe_per_step)
                out["roofline"]["whole_path_issued_TFLOPs"] = tot["issued"] / elapsed / 1e12
                out["roofline"]["whole_path_issued_frac"] = tot["issued"] / elapsed / 1e12 / peak
        if args.profile_all:
            model.dnn.profile_begin(1)
            step()
            torch.cuda.synchronize()
            table = model.dnn.profile_end()
            table.pop("_all_launches", None)
            tot = sum(v["ms"] for v in table.values())
            print(f"# per-op GPU time, one step ({tot:.2f} ms in kernels)", file=sys.stderr)
            import collections
            import re as _re
            lev = collections.defaultdict(lambda: [0, 0.0])
            for k, v in table.items():
                mres = _re.search(r"@(\d+)x(\d+)", k)
                key = mres.group(1) if mres else "other"
                lev[key][0] += v["launches"]
                lev[key][1] += v["ms"]
            for k in sorted(lev, key=lambda q: -lev[q][1]):
                print(f"# level H={k:6s} launches={lev[k][0]:5d} {lev[k][1]:8.2f} ms {100*lev[k][1]/tot:5.1f}%", file=sys.stderr)
            for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
                print(f"# {k:44s} n={v['launches']:4d} {v['ms']:9.3f} ms {100*v['ms']/tot:5.1f}% "
                      f"{tf:7.1f} TF/s {gbs:8.0f} GB/s", file=sys.stderr)
        if world == 1 and args.precision == "fp32" and args.solver == "euler" and not args.no_alt:
            # the optional operand modes of the same kernels, same workload, reported beside the exact-fp32 value
            ref_x = x.clone()
            alts = {}
            alt_first = {"fp32": x[:1].clone()}
            for mode in ("bf16x3", "bf16", "fp16"):
                model.dnn.set_precision(mode)
                for _ in range(2):                             # eager pass + graph capture
                    xm = step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    xm = step()
                torch.cuda.synchronize()
                dt_m = time.perf_counter() - t1
                err = float((xm - ref_x).abs().pow(2).sum().sqrt() / ref_x.abs().pow(2).sum().sqrt())
                alt_first[mode] = xm[:1].clone()
                alts[mode] = {"value": args.steps * B * T / dt_m, "unit": "frames/s", "ms_per_step": 1e3 * dt_m / args.steps,
                              "rel_l2_vs_fp32_mode": err,
                              "storage": "fp32 activations, split-bf16 operands in the 3x3 convs" if mode == "bf16x3"
                                         else f"{mode} activations in HBM, {mode} matrix-core operands, fp32 accumulate / GroupNorm"}
                # the mode's own dominant kernel (3x3 conv with fused GroupNorm input) against BOTH roofs
                model.dnn.profile_begin(0)
                step()
                torch.cuda.synchronize()
                pm = model.dnn.profile_end()
                dm = pm.get("dominant_conv3x3")
                if dm and dm["ms"] > 0:
                    tf = dm["issued"] / (dm["ms"] * 1e-3) / 1e12
                    gbs = dm["bytes"] / (dm["ms"] * 1e-3) / 1e9
                    alts[mode]["roofline"] = {
                        "kernel": ("flowse::conv3x3_halo_bf16_kernel (fp32 storage, split-bf16 x3 operands, LDS halo)" if mode == "bf16x3" else
                                   "flowse::conv3x3_pc16_kernel (+ conv3x3_halo_bf16_kernel on small launches): persistent "
                                   "producer/consumer LDS-halo 3x3 conv, 16-bit MFMA operands, fused GroupNorm+SiLU input"),
                        "bound": "hbm" if gbs / HBM_PEAK_GBS > tf / PEAK_16BIT_MATRIX_TFLOPS else "mfma",
                        "achieved_TFLOPs_issued": tf, "mfma_peak_TFLOPs": PEAK_16BIT_MATRIX_TFLOPS,
                        "mfma_frac": tf / PEAK_16BIT_MATRIX_TFLOPS,
                        "achieved_GBs_algorithmic": gbs, "hbm_peak_GBs": HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS,
                        "launches": dm["launches"], "avg_launch_ms": dm["ms"] / dm["launches"]}
                    tot = pm.get("_all_launches")
                    if tot:
                        alts[mode]["roofline"]["whole_path_issued_TFLOPs"] = tot["issued"] / (dt_m / args.steps) / 1e12
            model.dnn.set_precision("fp32")
            out["alt_precision"] = alts
            # single-utterance latency shape (the reference's own usage: one utterance per sampler call, evaluate.py:97)
            shapes = {}
            for (b1, t1) in ((1, 256),):
                Y1, Z1 = Y[:b1, :, :, :t1].contiguous(), Z[:b1, :, :, :t1].contiguous()

                def step1():
                    return get_white_box_solver("euler", model.ode, model, Y=Y1, Y_prior=Y1, T_rev=1.0, t_eps=0.03, N=NS,
                                                z=Z1)()[0]
                for _ in range(3):
                    step1()
                torch.cuda.synchronize()
                reps = max(args.steps, 5)
                t1s = time.perf_counter()
                for _ in range(reps):
                    step1()
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1s) / reps
                model.dnn.profile_begin(1)
                step1()
                torch.cuda.synchronize()
                pm = model.dnn.profile_end()
                shapes[f"[{b1},1,{F},{t1}]"] = {"value": b1 * t1 / dt1, "unit": "frames/s", "ms_per_step": 1e3 * dt1,
                                                 "ms_per_nfe": 1e3 * dt1 / NS,
                                                 "launches_per_nfe": pm["_all_launches"]["launches"] / NS}
            out["alt_shapes"] = shapes
            # one GPU's share of BASELINE config[3] at 8 GPUs (824 / 8 = 103 ragged utterances) through enhance_sharded
            a2 = argparse.Namespace(**vars(args))
            a2.utts, a2.steps, a2.warmup = VBDMD_UTTS, 1, 2
            vb = run_vbdmd(a2, model, dev, 1, 0, backend, False, share_of=(0, 8))
            out["alt_workloads"] = {"vbdmd_one_gpu_share": {k: vb[k] for k in ("value", "unit", "ms_per_step", "config",
                                                                              "frames_value_counts", "plan", "per_rank")}}
            out["alt_workloads"]["vbdmd_one_gpu_share"]["vs_headline_rate"] = vb["value"] / value
            out["alt_workloads"]["vbdmd_rank0_tail_probe"] = rank0_tail_probe(dev, 8, args.batch)
            if (B, T, NS) == (8, 256, 5) and not os.environ.get("FLOWSE_BENCH_NO_CONFIG4"):
                out["alt_workloads"]["config4_rk4_fp16"] = run_config4(model, dev)
        if vb_strong is not None:
            ref1 = n1_full_set_reference() if (args.utts, args.batch, NS, args.precision) == (VBDMD_UTTS, 8, 5, "fp32") else None
            vs = {k: vb_strong[k] for k in ("value", "unit", "ms_per_step", "scaling", "config", "frames_value_counts", "plan",
                                           "per_rank")}
            vs["n1_full_set_reference"] = ref1
            vs["vs_N1_full_set"] = (vb_strong["value"] / ref1["value"]) if ref1 else None
            out.setdefault("alt_workloads", {})["vbdmd_strong"] = vs
        if world == 1 and not args.no_cpu_baseline:
            want_oracle = args.solver == "euler" and T <= 256        # 1 utterance x N NFE of the CPU oracle: ~10 s at T = 256
            x_or, out["cpu_baseline"] = cpu_baseline(sd, NS, T, args.cpu_reps, oracle_utt=(rank * B) if want_oracle else None)
            out["gpu_vs_cpu"] = value / out["cpu_baseline"]["value"]
            if x_or is not None:
                # parity echo inside the bench line (the tests are the gate): utterance 0 of the timed batch, the full
                # N-step sampler, HIP path vs the pinned CPU oracle -- for `value`'s mode and for every alt mode
                def rel(a):
                    a = a.detach().cpu()
                    return float((a - x_or).abs().pow(2).sum().sqrt() / x_or.abs().pow(2).sum().sqrt())
                firsts = locals().get("alt_first") or {args.precision: x[:1]}
                out["rel_l2_vs_oracle"] = {"sample": f"utterance 0 of the batch, [1,1,{F},{T}], N={NS} euler, vs oracle/ "
                                                     "(torch-CPU fp32 restatement pinned to the reference's outputs); "
                                                     "north_star bar 1e-3 applies to fp32",
                                           args.precision: rel(firsts[args.precision])}
                for mode, xm0 in firsts.items():
                    if mode != args.precision:
                        out["rel_l2_vs_oracle"][mode] = rel(xm0)
                        if mode in out.get("alt_precision", {}):
                            out["alt_precision"][mode]["rel_l2_vs_oracle"] = out["rel_l2_vs_oracle"][mode]
        if args.solver != "euler" or (B, T) == (32, 1024):
            out["parity_note"] = ("fixed-step heun / rk4 have no reference counterpart (SURVEY 8 a19): their tableau is pinned "
                                  "against the oracle-VF composition at small size (tests/test_gpu_model.py); at this full "
                                  "size the run is property-checked only (finite, bit-identical repeat, batch-independent)")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
