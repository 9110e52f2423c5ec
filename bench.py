#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json).

metric   bases scanned / second: the PWM log-odds scan of a18 (both strands: scan + `> 0` threshold + ordered
         hit records, everything resident in HBM), on BASELINE configs[1]: 100k sequences x 200 bp against
         200 PWMs of length 12 per GPU.
step     one pass of that scan over the rank's shard (forward + reverse strand), followed, when N > 1, by the sum
         of the 2 x K hit histogram over the ranks (RCCL through the C ABI, on the scan's stream).
scaling  N = 1: BASELINE configs[1].  N > 1: the headline line is BASELINE configs[2] AS WRITTEN - the SAME 100k reads
         split evenly over the N ranks ("scaling": "strong"; `value` = 100k x 200 bases / the slowest rank's step);
         the weak-scaling run (every rank its own 100k x 200 bp shard) is reported beside it under "weak", and the
         split on whole 5000-read ordering batches under "strong.modes".

`python bench.py --gpus N` without a launcher starts the N rank processes itself (the parent never touches the
GPU); under `torch.distributed.run` (RANK / WORLD_SIZE set) it is one of the ranks.

One JSON line on stdout (rank 0).  `roofline` is measured with HIP events on the stream the kernels run on, inside
the timed region; `cpu_baseline` is the CPU port of the reference algorithm (Julia is not installed, so not the
reference itself) timed on a bounded sample on rank 0 at N == 1.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (no sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input matrix peak (= vector peak), v_mfma_f32_32x32x2_f32
# PMC traffic per (kernel, launch shape), written by tools/summarize_traffic.py from the passes of tools/profile_round.sh
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r05_scan_hbm_traffic.json", "r04_scan_hbm_traffic.json", "r03_scan_hbm_traffic.json")]
TRAIN_PMC_FILE = os.path.join(ROOT, "profiles", "r05_train_kernels_pmc.json")   # tools/summarize_train_pmc.py


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seqs", type=int, default=100_000, help="sequences per GPU (weak) / in total (strong)")
    ap.add_argument("--len", type=int, default=200)
    ap.add_argument("--pwms", type=int, default=200)
    ap.add_argument("--pwm-len", type=int, default=12)
    ap.add_argument("--cpu-sample", type=int, default=20_000, help="sequences in the CPU-baseline sample (vectorised port)")
    ap.add_argument("--cpu-literal-sample", type=int, default=300, help="sequences timed with the literal soft-float port")
    ap.add_argument("--train-groups", type=int, default=64, help="mini-batches (of 6 reads) per optimiser step per GPU")
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--filters", type=int, default=200)
    ap.add_argument("--filter-len", type=int, default=12)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed scan (+ train): skip dense / consumers / host-entry legs")
    ap.add_argument("--no-big", action="store_true", help="skip the configs[3] / configs[4] shard-shape legs")
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` with no launcher around it: start N rank processes of this script, one per GPU.  The
    parent initialises no GPU and replaces no process image; it waits and returns the worst exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:           # a dead rank leaves the others waiting in a collective
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            p.kill()
    return rc


def timed_region(step, steps, sync, barrier):
    barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    sync()
    barrier()
    sync()
    return time.perf_counter() - t0, out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # MOTIFS_BENCH_REHEARSE=1: every rank on device 0 over gloo — a rehearsal of the N > 1 control flow on a
    # one-GPU box (numbers meaningless); the real N > 1 run is one rank per GPU, RCCL over xGMI.
    rehearse = os.environ.get("MOTIFS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    watchdog = None
    if world > 1:
        # a rank that cannot finish its rendezvous or its communicator (another rank died, RCCL hangs on a link) must not sit in the job for ever:
        # past MOTIFS_BENCH_INIT_TIMEOUT seconds (default 240) without the set-up being complete the process leaves with status 5
        import threading

        def _give_up():
            sys.stderr.write(f"bench.py rank {rank}: process group / communicator not ready after the init timeout; exiting\n")
            sys.stderr.flush()
            os._exit(5)
        watchdog = threading.Timer(float(os.environ.get("MOTIFS_BENCH_INIT_TIMEOUT", "240")), _give_up)
        watchdog.daemon = True
        watchdog.start()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from _pkg import load_pkg

    pkg = load_pkg()
    lib, sy, par = pkg._lib, pkg.synth, pkg.parallel
    N, L, K, PL = args.seqs, args.len, args.pwms, args.pwm_len
    dev = torch.device("cuda", local_rank)

    def sync():
        torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    ctx = lib.Context(local_rank)
    # One ordinary (non-default) stream for everything: torch's allocations / fills run on it as its current stream, the
    # library's kernels and its RCCL collectives are enqueued on it through motifs_ctx_set_stream, so the stream itself
    # orders them (ABI 1 silently made a private stream here, and the two sides raced).
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    ctx.set_stream(work_stream.cuda_stream)
    # gpu_scan over many shards: a call returns once its hit totals are known and its records follow in stream order
    # (motifs_ctx_set_records_in_stream_order) - the host prepares step i + 1 under the record writes of step i; the timed region
    # ends with a device synchronize, so every record of every step is written inside it
    ctx.set_records_in_stream_order(True)
    reducer, reducer_note = par.make_reducer(ctx, prefer_rccl=not rehearse)
    if watchdog is not None:
        watchdog.cancel()
    if world > 1 and not rehearse and not isinstance(reducer, par.RcclReducer) and os.environ.get("MOTIFS_BENCH_ALLOW_HOST_SUMS") != "1":
        # make_reducer agreed on this over the process group (a MIN over the ranks): every rank takes this exit, none is left in a collective.
        # A line measured with host-staged sums would not be the RCCL-over-xGMI path BASELINE configs[2] names.
        sys.stderr.write(f"bench.py rank {rank}: {reducer_note}; refusing to report an N > 1 line without RCCL (MOTIFS_BENCH_ALLOW_HOST_SUMS=1 overrides)\n")
        sys.stderr.flush()
        dist.destroy_process_group()
        sys.exit(4)

    # ---- synthetic inputs ------------------------------------------------------
    seed = sy.SEED_BASE + 2
    pwms, lens = sy.gen_pwm_bank(K, seed, len_lo=PL, len_hi=PL, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)

    def make_shard(codes_np, n0):
        n = codes_np.shape[0]
        raw = torch.from_numpy(codes_np).to(dev)
        dcodes = torch.zeros(lib.Context.codes_bytes(n, L), dtype=torch.uint8, device=dev)
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, n, L, dcodes.data_ptr())
        need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), n, L, None, None, 0, n0=n0) if n else (0, 0)
        cap = int(max(need) * 1.05) + 1024
        hits = [torch.empty((cap, 3), dtype=torch.int32, device=dev) for _ in range(2)]
        hsc = [torch.empty(cap, dtype=torch.int16, device=dev) for _ in range(2)]
        counts = torch.zeros((2, K), dtype=torch.int64, device=dev)
        sync()
        return {"n": n, "n0": n0, "codes": dcodes, "need": need, "cap": cap, "hits": hits, "hsc": hsc, "counts": counts,
                "ptrs": (dcodes.data_ptr(), [h.data_ptr() for h in hits], [s.data_ptr() for s in hsc], counts.data_ptr())}

    def scan_step(sh):
        tot = 0
        if sh["n"]:
            # gpu_scan (_h3_1_alignment.jl:89-99): forward and reverse strand in one call, one host wait
            pc, ph, ps, pk = sh["ptrs"]
            tot = sum(ctx.pwm_scan_hits_both_dev(bank, lens, pc, sh["n"], L, ph, ps, sh["cap"], n0=sh["n0"], counts_ptr=pk))
        if world > 1:   # the one real exchange of the scan: the K int64 hit counts of both strands, queued on the
            reducer.hist_sum_(sh["counts"])  # scan's stream behind the kernels that wrote them (motifs_hist_allreduce); the next scan queues behind it
        return tot

    # weak: one 100k shard per rank
    codes = sy.gen_codes(N, L, seed + 1000 * rank, n_plant=5, k=PL)
    weak = make_shard(codes, rank * N)
    # The device comes out of the host-side set-up (data generation, uploads) at low clocks and needs ~25 steps (~35 ms)
    # of this workload to reach its steady state (tools: 1.52, 1.45, 1.41, 1.38, 1.36, 1.35 ms for successive groups of
    # five steps after an idle spell).  A fixed untimed pre-heat precedes the W warm-up steps so that the K timed steps
    # measure the steady state whatever W is.
    PREHEAT = 150      # (round 5: 40 left the first timed region ~2 % slower than the second whichever mode came first; steps 40-150 still speed up)
    # ... and the other clock, for the record: the first five steps after the set-up and half a second of idling, each
    # waited for on its own (what a caller who scans once sees)
    time.sleep(0.5)
    cold = []
    for _ in range(5):
        t0 = time.perf_counter()
        scan_step(weak)
        sync()
        cold.append((time.perf_counter() - t0) * 1e3)
    for _ in range(PREHEAT):
        scan_step(weak)
    for _ in range(args.warmup):
        scan_step(weak)
    sync()
    # HIP events inside the timed region only around the roofline kernel (the candidate filter): an event pair costs a few
    # microseconds of stream time per section; the other kernels are timed in an extra pass after the clock has stopped
    ctx.enable_timing(slots=[lib.KS_SCAN_COUNT])
    ctx.reset_timing()
    dt, nhits = timed_region(lambda: scan_step(weak), args.steps, sync, barrier)
    ctx.enable_timing(False)
    if world > 1:
        dt = float(par.host_all_reduce(torch.tensor([dt], dtype=torch.float64), dist.ReduceOp.MAX).item())
    tot_hits = torch.tensor([nhits], dtype=torch.int64)
    if world > 1:
        par.HostReducer(ctx).sum_i64_(tot_hits)

    kms = {"count": ctx.kernel_ms(lib.KS_SCAN_COUNT)}
    # the library's DEFAULT mode for the same K steps (every call waits for its records, as all rounds before the fourth measured it)
    ctx.set_records_in_stream_order(False)
    for _ in range(args.warmup):
        scan_step(weak)
    ctx.enable_timing(slots=[lib.KS_SCAN_COUNT])        # the same event pair per step as in the headline region
    dt_default, _ = timed_region(lambda: scan_step(weak), args.steps, sync, barrier)
    ctx.enable_timing(False)
    ctx.reset_timing()
    if world > 1:
        dt_default = float(par.host_all_reduce(torch.tensor([dt_default], dtype=torch.float64), dist.ReduceOp.MAX).item())
    ctx.set_records_in_stream_order(True)
    ctx.enable_timing(slots=[lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL])
    ctx.reset_timing()
    for _ in range(args.steps):
        scan_step(weak)
    sync()
    ctx.enable_timing(False)
    kms["offsets"] = ctx.kernel_ms(lib.KS_SCAN_OFFSETS)
    kms["fill"] = ctx.kernel_ms(lib.KS_SCAN_FILL)

    # ---- strong scaling: BASELINE configs[2] as written (the same N reads split over the ranks) -------------------
    strong = None
    strong_kms = None
    if world > 1:
        all_codes = sy.gen_codes(N, L, seed, n_plant=5, k=PL)          # the rank-0 shard of the weak run, on every rank
        modes = {}
        for mode, align in (("ordering_batches", lib.SCAN_BATCH), ("even", 1)):
            lo, hi = par.shard_range(N, rank, world, align=align)
            sh = make_shard(np.ascontiguousarray(all_codes[lo:hi]), lo)
            for _ in range(PREHEAT + args.warmup):
                scan_step(sh)
            sync()
            if mode == "even":
                ctx.enable_timing(slots=[lib.KS_SCAN_COUNT])
                ctx.reset_timing()
            sdt, sh_hits = timed_region(lambda: scan_step(sh), args.steps, sync, barrier)
            if mode == "even":
                ctx.enable_timing(False)
                strong_kms = ctx.kernel_ms(lib.KS_SCAN_COUNT)
            sdt = float(par.host_all_reduce(torch.tensor([sdt], dtype=torch.float64), dist.ReduceOp.MAX).item())
            sizes = [b - a for a, b in (par.shard_range(N, r, world, align=align) for r in range(world))]
            modes[mode] = {"value": float(N) * L * args.steps / sdt, "unit": "bases/s", "ms_per_step": sdt / args.steps * 1e3,
                           "shard_align": align, "shard_sizes": sizes, "hist_total_hits": int(sh["counts"].sum().item()),
                           "rank0_reads": hi - lo if rank == 0 else None}
            del sh
        strong = dict(modes["even"], scaling="strong", seqs_total=N, modes=modes,
                      note="BASELINE configs[2]: the same reads split over the ranks.  'even' (shard_align 1): 12 500 reads per rank at 8 ranks; "
                           "the concatenated records are in sequence-block-major order and give the single-device dictionaries of modify_w_found! "
                           "(_h3_1_alignment.jl:38-52).  'ordering_batches' (shard_align 5000): the concatenation is the single-device record list "
                           "bit for bit, but 20 batches over 8 ranks split 3,3,3,3,2,2,2,2.  The value quoted is the even mode's")
        del all_codes
    else:
        strong = {"value": float(N) * L * args.steps / dt, "unit": "bases/s", "ms_per_step": dt / args.steps * 1e3, "scaling": "strong",
                  "seqs_total": N, "shard_sizes": [N], "note": "N = 1: the same workload as the headline line"}

    extras = {}
    Lout = L - PL + 1
    hits, hsc, need, cap, dcodes = weak["hits"], weak["hsc"], weak["need"], weak["cap"], weak["codes"]
    # ---- what configs[2]'s shards cost on ONE GPU (N == 1): the step on 1/2, 1/4, 1/8 of the reads, with the histogram sum of a
    # one-rank RCCL communicator inside it.  8 ranks can be at most (full step) / (1/8 step) faster than one, whatever xGMI does:
    # the fixed costs of a step (launches, row scans, the host's wait, the collective's launch) do not shrink with the shard.
    comm1, red1 = None, None
    if world == 1 and not args.no_extras:
        try:
            comm1 = lib.Comm(ctx, lib.Comm.unique_id(), 1, 0)
            red1 = par.RcclReducer(comm1)
        except lib.MotifsError as e:
            extras["strong_proxy_note"] = f"one-rank RCCL communicator unavailable ({e}); shard steps timed without the collective"
        full_ms = dt / args.steps * 1e3
        proxy = {}
        for n_ in (N // 2, N // 4, N // 8):
            shp = make_shard(np.ascontiguousarray(codes[:n_]), 0)

            def pstep(shp=shp):
                pc, ph, ps, pk = shp["ptrs"]
                tot_ = sum(ctx.pwm_scan_hits_both_dev(bank, lens, pc, shp["n"], L, ph, ps, shp["cap"], n0=0, counts_ptr=pk))
                if red1 is not None:
                    red1.hist_sum_(shp["counts"])
                return tot_
            for _ in range(PREHEAT):
                pstep()
            pdt, _ = timed_region(pstep, args.steps, sync, barrier)
            pms = pdt / args.steps * 1e3
            proxy[str(n_)] = {"reads": n_, "ms_per_step": pms, "bases_per_s": n_ * L / (pms * 1e-3), "full_step_over_this": full_ms / pms,
                              "ideal": N / n_}
            del shp
        extras["strong_proxy"] = {
            "what": "BASELINE configs[2] priced on one GPU: the timed scan step on 1/2, 1/4 and 1/8 of the 100k reads (what a rank of 2 / 4 / 8 "
                    "scans), the 2 x K histogram summed by a one-rank RCCL communicator inside the step",
            "full_shard_ms_per_step": full_ms, "shards": proxy,
            "implied_speedup_bound_at_8_ranks": proxy[str(N // 8)]["full_step_over_this"],
            "north_star_needs": 6.0,
            "collective": "motifs_comm_allreduce_sum_i64_dev, one rank" if red1 is not None else "none",
        }
    if not args.no_extras and rank == 0:           # untimed side legs, all local to one device: rank 0 only
        # ---- the a17 kernel on its own: dense (K, nb, L-len+1) fp16 scores (untimed extra leg) ----
        # Reads per launch: the tensor of a launch is nb x 189 x 200 x 2 B, so the 100k reads go in slices.  The candidate
        # kernel hands 8 reads to a block and the chip holds 1024 of its blocks at a time, so a slice of 3 x 8192 = 24576
        # reads (1.86 GB) is three full rounds of blocks; the 20k-read slice of the earlier rounds (1.5 GB, 2.44 rounds: the
        # third one 44 % full) is timed beside it.
        # The clock state this leg finds the chip in moves a 5-launch average by +-8 % (the same binary, the same box, minutes
        # apart: 0.52-0.60 of the peak), so: 8 untimed launches, then 5 groups of 8 timed ones; the MEDIAN group is the figure and
        # the slowest and fastest groups are reported beside it.
        def _dense_rate(nb_):
            t = torch.empty((Lout, nb_, K), dtype=torch.int16, device=dev)
            for _ in range(8):
                ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), nb_, L, t.data_ptr(), Lout)
            by_ = nb_ * L + K * 4 * PL * 2 + nb_ * K * Lout * 2         # SURVEY §8d dense-score contract
            groups = []
            for _ in range(5):
                ctx.enable_timing(True)
                ctx.reset_timing()
                for _ in range(8):
                    ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), nb_, L, t.data_ptr(), Lout)
                ms_, n_ = ctx.kernel_ms(lib.KS_SCAN_DENSE)
                ctx.enable_timing(False)
                groups.append((ms_ / n_, by_ / (ms_ / n_ * 1e-3) / 1e9))
            groups.sort()
            ms_med, gbs_med = groups[len(groups) // 2]
            return t, ms_med, 1, gbs_med, [g[1] for g in groups]
        nb20 = min(N, 20_000)
        dense, ms20, n20, gbs20, grp20 = _dense_rate(nb20)
        del dense
        nb = min(N, 24_576)
        dense, dense_ms, dense_n, dense_gbs, dense_grp = _dense_rate(nb)
        # full-size check of the dense tensor against the hit records of the same reads (forward strand): the same
        # number of positive entries and the same sum of score bit patterns
        nrec = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), nb, L, 0, hits[0].data_ptr(), hsc[0].data_ptr(), cap, n0=rank * N)
        pos = dense > 0
        dense_pos = int(pos.sum().item())
        dense_sum = int(dense[pos].to(torch.int64).sum().item())
        rec_sum = int(hsc[0][:nrec].to(torch.int64).sum().item())
        assert dense_pos == nrec and dense_sum == rec_sum, f"dense tensor disagrees with the hit records: {dense_pos} vs {nrec}"
        del dense, pos

        # ---- the consumers of the records on the forward-strand records of this shard ----
        post = pkg.post
        nrec = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, hits[0].data_ptr(), hsc[0].data_ptr(), cap, n0=0)

        def _ms(fn, reps=3):
            fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return (time.perf_counter() - t0) / reps * 1e3
        mn, mx = post.score_range(ctx, hits[0], hsc[0], nrec, K)
        thr, _ = post.sweep_thresholds(mn, mx)
        thr_t = torch.from_numpy(np.ascontiguousarray(thr).view(np.int16)).to(dev)
        cnt_t = torch.zeros(thr.shape, dtype=torch.int64, device=dev)
        mn_t = torch.empty(K, dtype=torch.int16, device=dev)
        mx_t = torch.empty(K, dtype=torch.int16, device=dev)
        th_t = torch.from_numpy(np.ascontiguousarray(thr[:, thr.shape[1] // 2]).view(np.int16)).to(dev)
        oh_t, os_t = torch.empty_like(hits[0]), torch.empty_like(hsc[0])
        cm_t = torch.zeros((K, int(lens.max()), 4), dtype=torch.int32, device=dev)
        extras["record_consumers"] = {
            "records": int(nrec),
            "score_range_ms": _ms(lambda: ctx.hits_minmax_dev(hits[0].data_ptr(), hsc[0].data_ptr(), nrec, K, mn_t.data_ptr(), mx_t.data_ptr())),
            "threshold_sweep_ms": _ms(lambda: ctx.hits_threshold_counts_dev(hits[0].data_ptr(), hsc[0].data_ptr(), nrec, K, thr_t.data_ptr(),
                                                                             thr.shape[1], cnt_t.data_ptr())),
            "threshold_sweep_steps": int(thr.shape[1]),
            "filter_ms": _ms(lambda: ctx.hits_filter_dev(hits[0].data_ptr(), hsc[0].data_ptr(), nrec, K, th_t.data_ptr(), oh_t.data_ptr(),
                                                          os_t.data_ptr())),
            "count_matrices_ms": _ms(lambda: ctx.hits_count_matrices_dev(hits[0].data_ptr(), nrec, dcodes.data_ptr(), L, 0, lens, K,
                                                                          int(lens.max()), 0, cm_t.data_ptr())),
            "note": "SURVEY 8f rows 2-3 on the forward-strand records of the shard, wall time per call incl. launch and sync",
        }
        del oh_t, os_t

        # ---- attainable HBM rates in this run: a device copy and a fill of ~1 GB, torch kernels ----
        def _rate(fn, nbytes, reps=10):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            sync()
            return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
        hx = torch.empty(1 << 29, dtype=torch.int16, device=dev)
        hy = torch.empty_like(hx)
        hbm_fill_gbs = _rate(lambda: hx.zero_(), hx.numel() * 2)
        hbm_copy_gbs = _rate(lambda: hy.copy_(hx), 2 * hx.numel() * 2)
        del hx, hy
        extras["dense_kernel"] = {
            "kernel": "scan_dense_fused<3,7,8> (a17 greedy_search! drop-in in one kernel: MFMA tiles of 32 consecutive reads at one start, candidates re-scored from LDS, "
                      "the contiguous 32 x K span streamed out of an LDS window: (K,nb,L-len+1) fp16, zeros + exact scores of the hits, every byte once; "
                      "banks past 256 PWMs or 20 positions: scan_cand_kernel_q + stage_hits<12,.,2>)",
            "checked": "positive entries == hit records of the same reads, score checksums equal",
            "bound": "hbm", "achieved": dense_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dense_gbs / HBM_PEAK_GBS, "frac_of_measured_fill": dense_gbs / hbm_fill_gbs,
            "avg_launch_ms": dense_ms / dense_n, "seqs_per_launch": nb,
            "timing": "8 untimed launches, 5 groups of 8 timed (HIP events around each launch); the median group is quoted",
            "groups_gbs_fastest_to_slowest": dense_grp,
            "bases_per_s_one_strand": nb * L / (dense_ms / dense_n * 1e-3),
            "launch_size_note": "24576 reads = 768 blocks of 32 reads x 2 halves of the starts = six rounds of one block per CU",
            "at_20k_reads_per_launch": {"achieved": gbs20, "frac": gbs20 / HBM_PEAK_GBS, "avg_launch_ms": ms20 / n20, "seqs_per_launch": nb20,
                                        "groups_gbs_fastest_to_slowest": grp20},
        }
        extras["hbm_measured"] = {"fill_gbs": hbm_fill_gbs, "copy_gbs_read_plus_write": hbm_copy_gbs,
                                  "note": "torch zero_() / copy_() of 1 GiB in this run; nominal peak 8000 GB/s"}

        # ---- the entry a Julia `ccall` binds: host f32 one-hot in, host records out (_h3_1_alignment.jl:57-87, x2 strands) ----
        if rank == 0:
            onehot = sy.codes_to_onehot(codes)                          # (N, 4L) f32 = the bytes of data.data_matrix (4L,1,N)
            cap_h = int(max(need))
            ctx.pwm_scan(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, False, cap=cap_h)      # warm-up (buffers, bank cache)
            t0 = time.perf_counter()
            n_h = 0
            for rc in (False, True):
                f, _ = ctx.pwm_scan(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, rc, cap=cap_h)
                n_h += len(f)
            hdt = time.perf_counter() - t0
            assert n_h == sum(need), (n_h, need)
            ctx.pwm_scan_both(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, cap=cap_h)
            t0 = time.perf_counter()
            both = ctx.pwm_scan_both(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, cap=cap_h)
            bdt = time.perf_counter() - t0
            assert len(both[0][0]) + len(both[1][0]) == sum(need)
            extras["host_entry"] = {
                "entry": "motifs_pwm_scan_both (gpu_scan, _h3_1_alignment.jl:89-99: host Float32 one-hot matrix in, the two strands' host records out, one upload)",
                "bases_per_s": N * L / bdt, "ms_both_strands": bdt * 1e3,
                "per_strand_entry": {"entry": "motifs_pwm_scan, one call per strand (get_pos_scores_arr, :57-87)", "bases_per_s": N * L / hdt,
                                     "ms_both_strands": hdt * 1e3},
                "host_matrix_bytes": int(onehot.nbytes), "h2d_bytes": int(lib.Context.codes_bytes(N, L)), "d2h_bytes_both_strands": int(n_h * 14),
                "note": "PCIe-inclusive, pageable host memory: the 16 B/base one-hot matrix becomes 1 B/base code rows on host threads (pinned) before the upload; 14 B/hit come down through a pinned ring; never `value`",
            }
            del onehot
            # ---- FASTA text -> base codes (loadfasta/helpers.jl:83-139), host threads only ----
            import tempfile

            rows = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
            with tempfile.NamedTemporaryFile("wb", suffix=".fa", delete=False) as fh:
                fa_path = fh.name
                for i in range(N):
                    fh.write(b">s%d\n" % i)
                    fh.write(rows[i].tobytes())
                    fh.write(b"\n")
            fa_bytes = os.path.getsize(fa_path)
            lib.fasta_read(fa_path)
            t0 = time.perf_counter()
            fa = lib.fasta_read(fa_path)
            fdt = time.perf_counter() - t0
            os.remove(fa_path)
            assert np.array_equal(fa, codes)
            extras["fasta_reader"] = {"bases_per_s": N * L / fdt, "file_MB_per_s": fa_bytes / fdt / 1e6, "ms": fdt * 1e3, "file_bytes": fa_bytes,
                                      "host_threads": min(32, os.cpu_count() or 1),
                                      "note": "motifs_fasta_read, query + fill call (the file is parsed twice), page-cache resident file"}

    # ---- BASELINE configs[3] and configs[4] at the shape of ONE rank's shard of the 8-GPU job (untimed side legs, rank 0) ----
    if not args.no_extras and rank == 0 and not args.no_big:

        def shard_leg(tag, n_, L_, K_, len_lo, len_hi, ws_limit, note, reps_=3):
            pw, ln = sy.gen_pwm_bank(K_, seed + 7, len_lo=len_lo, len_hi=len_hi, alpha=0.3)
            bk = sy.pad_bank(pw, ln)
            cd = sy.gen_codes(n_, L_, seed + 31, n_plant=5, k=len_hi)
            raw_ = torch.from_numpy(cd).to(dev)
            dc_ = torch.zeros(lib.Context.codes_bytes(n_, L_), dtype=torch.uint8, device=dev)
            ctx.encode_dev(raw_.data_ptr(), lib.DATA_CODES_U8, n_, L_, dc_.data_ptr())
            del raw_, cd
            torch.cuda.empty_cache()
            sync()
            free0 = torch.cuda.mem_get_info(dev)[0]
            ctx.set_workspace_limit(ws_limit)
            need_ = ctx.pwm_scan_hits_both_dev(bk, ln, dc_.data_ptr(), n_, L_, None, None, 0)
            cap_ = int(max(need_)) + 1024
            h_ = [torch.empty((cap_, 3), dtype=torch.int32, device=dev) for _ in range(2)]
            s_ = [torch.empty(cap_, dtype=torch.int16, device=dev) for _ in range(2)]
            k_ = torch.zeros((2, K_), dtype=torch.int64, device=dev)

            def one():
                return ctx.pwm_scan_hits_both_dev(bk, ln, dc_.data_ptr(), n_, L_, [x.data_ptr() for x in h_], [x.data_ptr() for x in s_], cap_,
                                                  counts_ptr=k_.data_ptr())
            for _ in range(2):
                got_ = one()
            plan_ = ctx.scan_plan()
            sync()
            peak_bytes = free0 - torch.cuda.mem_get_info(dev)[0]               # records + the library's workspaces (not torch's: hipMalloc)
            ctx.enable_timing(slots=[lib.KS_SCAN_COUNT])
            ctx.reset_timing()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps_):
                got_ = one()
            sync()
            leg_dt = (time.perf_counter() - t0) / reps_
            cms, cn = ctx.kernel_ms(lib.KS_SCAN_COUNT)
            # the other two stages, timed in a pass of their own (an event pair per section costs stream time)
            ctx.enable_timing(slots=[lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL])
            ctx.reset_timing()
            for _ in range(reps_):
                one()
            sync()
            oms, on_ = ctx.kernel_ms(lib.KS_SCAN_OFFSETS)
            fms, fn_ = ctx.kernel_ms(lib.KS_SCAN_FILL)
            ctx.enable_timing(False)
            ctx.set_workspace_limit(0)
            assert tuple(got_) == tuple(need_), (got_, need_)
            assert int(k_.sum().item()) == sum(got_)                           # histogram == records, both strands
            # records of every ordering batch in findall order (l slowest) - a full-size, size-independent property (in chunks: the
            # int64 temporaries of 2e9 records at once would not fit beside them)
            last_ = None
            for c0 in range(0, got_[0], 1 << 27):
                f0 = h_[0][c0: min(got_[0], c0 + (1 << 27))]
                bq_ = torch.div(f0[:, 1] - 1, lib.SCAN_BATCH, rounding_mode="floor").to(torch.int64)
                key_ = (bq_ * (L_ + 1) + f0[:, 2].to(torch.int64)) * lib.SCAN_BATCH + (f0[:, 1].to(torch.int64) - 1) % lib.SCAN_BATCH
                key_ = key_ * (K_ + 1) + f0[:, 0].to(torch.int64)
                assert bool((key_[1:] > key_[:-1]).all()), "records are not in the reference's order"
                assert last_ is None or int(key_[0]) > last_, "records are not in the reference's order"
                last_ = int(key_[-1])
                del key_, bq_, f0
            windows = float(n_) * float(np.sum(L_ - ln + 1))                   # (PWM, start) pairs per strand
            flops = 2.0 * 4.0 * float(n_) * float(np.sum((L_ - ln + 1) * ln))  # 2 * 4 * len_k flop per pair (GEMM form)
            per_launch_ms = cms / max(cn, 1)
            launches_per_strand = cn / (2 * reps_)
            tfl = flops / (cms / (2 * reps_) * 1e-3) / 1e12                    # over all candidate launches of one strand
            stage_ms = {"scan_cand": cms / reps_, "stage_hits_row_scan": oms / reps_, "emit_records": fms / reps_}
            hits_strand = sum(got_) / 2.0
            lenp_ = (int(ln.max()) + 3) // 4 * 4
            # the other two stages against what bounds them.  Re-scoring (stage_hits / stage_hits_cg): VALU issue (1024 SIMDs, one wave-instruction
            # per 4 cycles, 2.4 GHz).  The ALGORITHMIC instruction count is the exact score alone, from the ISA of exact_score<LEN, true>: per 64
            # candidates lenp/4 + 1 shifts (the window's dwords doubled), lenp/4 byte alignments, lenp address adds (SDWA) and lenp - 1 binary16
            # adds - 30 wave-instructions at 12 positions, 50 at 20 - with the lenp LDS gathers beside them on the LDS pipe.  `frac` = hits x that /
            # the issue rate; what the kernel really issues per hit (PMC, profiles/r05_cg_stage_valu.json: the walk over the candidate entries, the
            # queue, validity tests, staging) is reported beside it as `issue_utilisation` - a kernel with twice the instructions would score the
            # same there, which is why it is not the roofline figure.
            valu_peak = 1024 * 2.4e9 / 4.0
            vf = os.path.join(ROOT, "profiles", "r05_cg_stage_valu.json")
            ipw, vsrc = 3.9, None
            if os.path.exists(vf):
                with open(vf) as fh:
                    vj = json.load(fh)
                ipw, vsrc = vj["valu_wave_insts_per_hit"], {"file": os.path.relpath(vf, ROOT), "commit": vj.get("commit"), "kernel": vj.get("kernel")}
            hits_per_s = hits_strand / (oms / (2 * reps_) * 1e-3)
            alg_ipc = (lenp_ // 4 + 1 + lenp_ // 4 + lenp_ + lenp_ - 1) / 64.0       # wave-instructions per candidate, exact score only
            rescoring = {"kernel": ("stage_hits_cg (chunk groups: one group's table slice in LDS per block) + row scan" if plan_["cg_chunks"] else
                                    "stage_hits (whole table in the LDS of one 16-wave block per CU when it is past 64 KB) + row scan"),
                         "bound": "valu-issue", "achieved": hits_per_s * alg_ipc / 1e9, "peak": valu_peak / 1e9, "unit": "G wave-instructions/s",
                         "frac": hits_per_s * alg_ipc / valu_peak, "algorithmic_wave_insts_per_candidate": alg_ipc,
                         "issue_utilisation": hits_per_s * ipw / valu_peak, "valu_wave_insts_per_hit_by_pmc": ipw, "insts_source": vsrc, "hits_per_s": hits_per_s,
                         "lds_gather_frac_of_lds_rate": hits_strand * lenp_ / (oms / (2 * reps_) * 1e-3) / (64.0 * 256 * 2.4e9),
                         "ms_per_strand": oms / (2 * reps_),
                         "note": "frac: the exact scores alone against the VALU issue rate (hits stand for candidates: 95 % of them are hits); issue_utilisation: every "
                                 "vector instruction the kernel issues (PMC pass of the chunk-group kernel at the configs[4] bank shape; the 16-wave form of configs[3] "
                                 "issues about the same per candidate but walks 15x more empty cells per hit)"}
            rec_bytes = hits_strand * 18.0
            records = {"kernel": "emit_records_cg" if plan_["cg_chunks"] else "emit_records", "bound": "hbm",
                       "achieved": rec_bytes / (fms / (2 * reps_) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": rec_bytes / (fms / (2 * reps_) * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_strand": rec_bytes,
                       "ms_per_strand": fms / (2 * reps_)}
            cand_roof = {"kernel": "scan_cand_kernel_* (the launches of one strand pass, summed)", "bound": "mfma", "achieved": tfl,
                         "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / MFMA_F16_PEAK_TFLOPS, "flops_per_strand": flops,
                         "avg_launch_ms": per_launch_ms, "traffic": None}
            largest = max(stage_ms, key=stage_ms.get)
            out_ = {
                "workload": note, "seqs": n_, "seq_len": L_, "pwms": K_, "pwm_len": [int(ln.min()), int(ln.max())],
                "ms_per_step_both_strands": leg_dt * 1e3, "bases_per_s": n_ * L_ / leg_dt, "hits_per_step": int(sum(got_)),
                "record_bytes_per_step": int(sum(got_)) * 14, "device_bytes_held": int(peak_bytes),
                "pwm_window_pairs_per_strand": windows, "workspace_limit_bytes": int(ws_limit),
                "candidate_launches_per_strand": launches_per_strand, "plan": plan_,
                "kernel_ms_per_step": stage_ms, "largest_stage_by_time": largest,
                "roofline": {"scan_cand": cand_roof, "stage_hits_row_scan": rescoring, "emit_records": records}[largest],
                "rooflines": {"scan_cand": cand_roof, "stage_hits_row_scan": rescoring, "emit_records": records},
                "checked": "both strands: totals == the count-only pass, histogram == records, forward records strictly ascending in "
                           "(batch, l, n, m): the reference's order",
            }
            del h_, s_, dc_, k_
            torch.cuda.empty_cache()
            return out_
        extras["cfg3_shard"] = shard_leg("cfg3", 62_500, 500, 512, 20, 20, 0,
                                         "BASELINE configs[3], one rank's shard of 8: 62 500 seqs x 500 bp vs 512 PWMs len 20, both strands, ordered records")
        extras["cfg4_shard"] = shard_leg("cfg4", 25_000, 1000, 2048, 8, 20, 4 << 30,
                                         "BASELINE configs[4] at a fifth of one rank's shard of 8: 25 000 seqs x 1000 bp vs 2048 PWMs len 8-20 (mixed), both strands, "
                                         "ordered records; workspace bound 4 GiB so that a strand crosses several super-batch launches")
        extras["cfg4_rank_shard"] = shard_leg("cfg4r", 125_000, 1000, 2048, 8, 20, 0,
                                              "BASELINE configs[4], ONE rank's whole shard of the 8-GPU job: 125 000 seqs x 1000 bp vs 2048 PWMs len 8-20, both strands, "
                                              "ordered records (~2e9 per strand, 55 GB), default 8 GiB workspace", reps_=2)

    # ---- conv-train step (BASELINE metric, second half): unrolled-ADMM forward/backward + AdaBelief ----
    train = None
    if not args.no_train:
        md = pkg.model
        hp = md.Hyperparam(filter_len=args.filter_len, M=args.filters)
        Gt = args.train_groups
        St = Gt * hp.batch_size
        cdl = md.ucdl(hp, L, ctx=ctx, seed=seed, arena_bytes=int((0.3 * Gt + 2) * (1 << 30)))   # a step of Gt mini-batches peaks at 0.17 GiB per mini-batch (arena_peak_bytes)
        tcodes = sy.gen_codes(St, L, seed + 77 + 1000 * rank, n_plant=5, k=args.filter_len)
        traw = torch.from_numpy(tcodes).to(dev)
        tdev = torch.zeros(lib.Context.codes_bytes(St, L), dtype=torch.uint8, device=dev)
        ctx.encode_dev(traw.data_ptr(), lib.DATA_CODES_U8, St, L, tdev.data_ptr())
        tloss = torch.zeros(Gt, dtype=torch.float32, device=dev)
        tgrad = torch.zeros(cdl.model.nP, dtype=torch.float32, device=dev)

        def tstep(g):
            par.dp_train_step(cdl.model, tdev.data_ptr(), g, tloss, tgrad, g * world, reducer=reducer)
        tstep(Gt)                                                   # warm-up
        sync()
        ctx.enable_timing(True)
        ctx.reset_timing()
        tdt, _ = timed_region(lambda: tstep(Gt), args.train_steps, sync, barrier)
        ctx.enable_timing(False)
        if world > 1:
            tdt = float(par.host_all_reduce(torch.tensor([tdt], dtype=torch.float64), dist.ReduceOp.MAX).item())
        gms, gn = ctx.kernel_ms(lib.KS_TRAIN_STEP)
        ista_ms, ista_n = ctx.kernel_ms(lib.KS_TRAIN_ISTA_BWD)      # event pairs around every launch of the step's largest kernel, inside the timed region
        # a4 on its own (SURVEY 8d "conv forward scan"): in L bytes of codes per read, out 2*c*M*4 bytes of dense codes
        a4_ms = cdl.model.time_filter_scan(tdev.data_ptr(), Gt, reps=10)
        c_rows = L - args.filter_len + 1
        a4_bytes = St * (L + 2 * c_rows * args.filters * 4)
        # ... and on a launch large enough to show what the kernel can do (code retrieval runs it over all reads): 683 mini-batches = 4 098 reads
        Gbig = 683
        big_codes = sy.gen_codes(Gbig * hp.batch_size, L, seed + 78, n_plant=5, k=args.filter_len)
        braw = torch.from_numpy(big_codes).to(dev)
        bdev = torch.zeros(lib.Context.codes_bytes(Gbig * hp.batch_size, L), dtype=torch.uint8, device=dev)
        ctx.encode_dev(braw.data_ptr(), lib.DATA_CODES_U8, Gbig * hp.batch_size, L, bdev.data_ptr())
        a4_big_ms = cdl.model.time_filter_scan(bdev.data_ptr(), Gbig, reps=10)
        a4_big_bytes = Gbig * hp.batch_size * (L + 2 * c_rows * args.filters * 4)
        del braw, bdev
        # a16 code retrieval (_1_code_retrieval.jl:33-56) as discover_motifs calls it: host codes in, host records out, mini-batches in file order
        n_ret = min(N, 30_000)
        cdl.model.retrieve_codes(codes[:n_ret], lib.DATA_CODES_U8, n_ret)
        t0 = time.perf_counter()
        ret_rec = cdl.model.retrieve_codes(codes[:n_ret], lib.DATA_CODES_U8, n_ret)
        ret_dt = time.perf_counter() - t0
        # a7's dense contraction on its own ("MFMA fraction is computed on the syntax-layer GEMM flops", f32 matrix peak)
        a7_ms = cdl.model.time_syntax_conv(tdev.data_ptr(), Gt, reps=10)
        l_rows = c_rows - hp.h + 1
        a7_flops = 2.0 * St * l_rows * hp.K * hp.h * 2 * args.filters
        # the reference's own schedule: one optimiser step per 6-read mini-batch (train.jl:40-46)
        for _ in range(3):
            tstep(1)
        g1_steps = 20
        g1dt, _ = timed_region(lambda: tstep(1), g1_steps, sync, barrier)
        # ... and 8 mini-batches per step: one rank's share of the 64-mini-batch step at 8 ranks (the strong-scaling price of the
        # train step on one GPU; with the gradient sum of a one-rank RCCL communicator when there is one)
        red8 = red1 if (world == 1 and red1 is not None) else reducer

        def tstep8():
            par.dp_train_step(cdl.model, tdev.data_ptr(), 8, tloss, tgrad, 8 * world, reducer=red8)
        for _ in range(3):
            tstep8()
        g8dt, _ = timed_region(tstep8, g1_steps, sync, barrier)
        # counter evidence of the step (PMC passes over the same 64-mini-batch step, tools/profile_r04.sh train): HBM bytes of the roofline
        # kernel per launch, and of the whole step against its time here
        a7_traffic, step_hbm = None, None
        if os.path.exists(TRAIN_PMC_FILE) and (Gt, L, args.filters, args.filter_len) == (64, 200, 200, 12):
            with open(TRAIN_PMC_FILE) as fh:
                tp = json.load(fh)
            for e in tp["kernels"]:
                if e["kernel"].startswith("k_ana_lds") and e["blocks"] > 1000:
                    a7_traffic = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / e["launches"]
            step_hbm = {"bound": "hbm", "bytes_per_step_by_pmc": tp["step"]["hbm_bytes"], "achieved": tp["step"]["hbm_bytes"] / (tdt / args.train_steps) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tp["step"]["hbm_bytes"] / (tdt / args.train_steps) / 1e9 / HBM_PEAK_GBS,
                        "launches_per_step": tp["step"]["launches"], "source": {"file": os.path.relpath(TRAIN_PMC_FILE, ROOT), "commit": tp.get("commit")},
                        "note": "counter bytes of one step (2 x FETCH_SIZE + WRITE_SIZE over every kernel of the step) / the step time measured here"}
        train = {
            "workload": f"unrolled-ADMM sparse coding, {Gt} mini-batches x {hp.batch_size} reads x {L} bp per GPU per "
                        f"optimiser step, {args.filters} filters len {args.filter_len}, h=12 K=24 q=32, 6+3 passes; f32 storage and elementwise "
                        "arithmetic, the four GEMMs of a large step as f16x3 matrix products (each float32 operand split into two binary16 numbers, three "
                        "v_mfma_f32_32x32x16_f16 per term, f32 accumulation: 22 mantissa bits)",
            "ms_per_step": tdt / args.train_steps * 1e3,
            "seqs_per_s": St * world * args.train_steps / tdt,
            "bases_per_s": St * world * L * args.train_steps / tdt,
            "device_ms_fwd_bwd": gms / max(gn, 1),
            "ms_per_step_g1": g1dt / g1_steps * 1e3,
            "seqs_per_s_g1": hp.batch_size * world * g1_steps / g1dt,
            "ms_per_step_g8": g8dt / g1_steps * 1e3,
            "g8_note": f"8 mini-batches per step per GPU (one rank's share of the {Gt}-mini-batch step at 8 ranks; gradient sum: "
                       f"{getattr(red8, 'kind', 'none')}): 8 ranks can be at most (ms_per_step / ms_per_step_g8) faster than one",
            "implied_speedup_bound_at_8_ranks": (tdt / args.train_steps) / (g8dt / g1_steps) if Gt == 64 else None,
            "g1_note": "reference schedule (train.jl:40-46): one AdaBelief step per mini-batch of 6 reads per GPU; ~270 launches of 2-10 us "
                       "(profiles/r03_g1_step_kernels.txt, DESIGN 3)",
            "filter_scan_a4": {
                "kernel": "k_onehot_bank_scan (warmup_ZY's conv(S,D) pair, model.jl:171-173: S is one-hot, so each output is fl bank rows picked by base "
                          "code from LDS and added in the GEMM's order; MOTIFS_NO_ONEHOT_SCAN=1: k_toep_wide, the Toeplitz GEMM on the f32 image)",
                "bound": "hbm", "achieved": a4_bytes / (a4_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a4_bytes / (a4_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": a4_ms, "algorithmic_bytes": a4_bytes,
                "reads_per_launch": St, "bases_per_s": St * L / (a4_ms * 1e-3),
                "at_4098_reads_per_launch": {"achieved": a4_big_bytes / (a4_big_ms * 1e-3) / 1e9, "frac": a4_big_bytes / (a4_big_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "avg_launch_ms": a4_big_ms, "algorithmic_bytes": a4_big_bytes, "reads_per_launch": Gbig * hp.batch_size},
            },
            "code_retrieval": {"reads_per_s": n_ret / ret_dt, "reads": n_ret, "records": int(len(ret_rec)), "ms": ret_dt * 1e3,
                               "entry": "motifs_model_retrieve_codes (a16, _1_code_retrieval.jl:33-56): host base codes in, host stored_code_component_t records out, "
                                        "wall time of one call incl. upload, forward passes and download"},
            "roofline": None,            # filled below: the kernel of the step that is largest by time
            "syntax_gemm_a7": (lambda f16x3: {
                "kernel": ("k_ana_f16x3 (a7: conv(ZY, F, flipped=true), model.jl:214,251, on v_mfma_f32_32x32x16_f16 with THREE products per term - every float32 operand "
                           "split into two binary16 numbers, 22 bits - + the pass that finds the image's largest magnitude; MOTIFS_GEMM_F32=1: k_ana_lds on "
                           "v_mfma_f32_32x32x2_f32)") if f16x3 else
                          "k_ana_lds (a7: conv(ZY, F, flipped=true), model.jl:214,251, on v_mfma_f32_32x32x2_f32; rows = reads x l, columns = K, reduction = h * 2M)",
                "bound": "mfma", "achieved": a7_flops / (a7_ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS / 3.0 if f16x3 else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": a7_flops / (a7_ms * 1e-3) / 1e12 / (MFMA_F16_PEAK_TFLOPS / 3.0 if f16x3 else MFMA_F32_PEAK_TFLOPS),
                "frac_of_f32_matrix_peak": a7_flops / (a7_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "avg_launch_ms": a7_ms, "flops_per_launch": a7_flops, "reads_per_launch": St, "traffic": a7_traffic if not f16x3 else None,
                "note": "algorithmic flops 2 * l * K * h * 2M per read (SURVEY 8d; the kernel pads K 24 -> 32 and l 178 -> 192).  f16x3: the peak is the dense f16 "
                        "peak over the three matrix instructions a term costs; the kernel is no longer bound by the matrix pipe but by the filter fragments it "
                        "pulls from L2 (437 MB per launch at three row tiles per fragment) and the 171 MB image from HBM",
            })(os.environ.get("MOTIFS_GEMM_F32") is None and St * ((l_rows + 31) // 32) >= 1024 and (2 * args.filters) % 16 == 0 and hp.h == 12),
            "step_hbm": step_hbm,
            "arena_peak_bytes": cdl.model.arena_peak(),
            "arena_peak_GiB_per_mini_batch": cdl.model.arena_peak() / Gt / 2**30,
            "loss_first_group": float(tloss[0].item()),
            "grad_allreduce_floats": int(cdl.model.nP) if world > 1 else 0,
            "reference_schedule_equivalent": f"{Gt * world} reference steps (batch 6) worth of reads per step",
            "graph": ("the reference's graph with its common subexpressions formed once (the syntax-layer synthesis with an unchanged "
                      "bank) and ADMM_DF's residuals telescoped (R_1 = 0, R_t = -theta_{t-2}; identical values, "
                      "tests/test_model_gpu.py::test_df_telescoping_equals_literal_sequence); MOTIFS_DF_LITERAL=1 runs the literal "
                      "sequence"),
        }
        # the step's largest kernel by time (profiles/r05_train_kernels_pmc.json: 5 launches, 12 % of the step): the VJP of update_ZY's fused ISTA
        # step.  Elementwise over the code images [reads][c][2M] f32: eight read (the two incoming gradients, the gradient of the combination formed
        # after the step, the step's output, ZY, the D-layer gradient image, FX, the duals) and four written (gradients of ZY, g1, FX, duals); the first
        # pass has no duals (ten streams), counted as twelve here, so `achieved` is an upper bound of at most 3 %
        img_bytes = float(St) * c_rows * 2 * args.filters * 4
        ista_launch_ms = ista_ms / max(ista_n, 1)
        ista_traffic = None
        if os.path.exists(TRAIN_PMC_FILE) and (Gt, L, args.filters, args.filter_len) == (64, 200, 200, 12):
            with open(TRAIN_PMC_FILE) as fh:
                for e in json.load(fh)["kernels"]:
                    if e["kernel"].startswith("k_zy_step2_bwd"):
                        ista_traffic = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / e["launches"]
        train["roofline"] = {
            "kernel": "k_zy_step2_bwd<8> (VJP of update_ZY's ISTA step relu(ZY - step (grad + penalty (ZY - FX - dual)) - step lambda), model.jl:237-245, with the "
                      "VJP of the combination formed after it; elementwise over twelve image-sized streams, 32 bytes per lane and stream)",
            "why_this_kernel": "largest kernel of the step by time", "share_of_step_time": ista_ms / args.train_steps / (tdt / args.train_steps * 1e3) if ista_n else None,
            "bound": "hbm", "achieved": 12 * img_bytes / (ista_launch_ms * 1e-3) / 1e9 if ista_n else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": 12 * img_bytes / (ista_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ista_n else None,
            "algorithmic_bytes_per_launch": 12 * img_bytes, "avg_launch_ms": ista_launch_ms, "launches_per_step": ista_n / max(args.train_steps, 1),
            "traffic": ista_traffic, "timing": "HIP event pairs around each launch on the context's stream, inside the timed region (MOTIFS_KS_TRAIN_ISTA_BWD)",
        }
        cdl.model.close()

    if isinstance(reducer, par.RcclReducer):       # the library's own communicator goes before torch's
        ctx.synchronize()
        reducer.comm.close()
    if comm1 is not None:
        ctx.synchronize()
        comm1.close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    bases = float(N) * L * world
    value = bases * args.steps / dt
    weak_line = {"value": value, "unit": "bases/s", "ms_per_step": dt / args.steps * 1e3, "scaling": "weak", "seqs_per_gpu": N,
                 "note": "every rank its own 100k x 200 bp shard (per-GPU work fixed as N grows)"}
    # One strand pass = scan_cand_kernel (the GEMM of the PWM bank with the one-hot windows on the matrix
    # cores: every window of every PWM, thresholded at -eps_k into 128-bit candidate cells) -> stage_hits
    # (exact binary16 re-scoring of the ~1 % candidates, staged hit words, row counts) -> row scan ->
    # emit_records ((m, n, l) + score records in the reference's order).
    # Dominant kernel: scan_cand_kernel, bound by the matrix cores.  Algorithmic flops per launch (SURVEY
    # 8d, GEMM form): 2 * 4*len * K flop per window, N * (L - len + 1) windows per strand launch.
    # The whole pass is also quoted against HBM on the fused-hits contract (N*L codes in, 14 B per hit out).
    hits_per_pass = nhits / 2.0
    n_launch = max(kms["count"][1], 1)
    strands_per_launch = 2.0 * args.steps / n_launch           # 2: one candidate launch takes both strands' banks (gpu_scan); 1: one per strand
    cand_ms = kms["count"][0] / n_launch
    head_reads = N                                              # reads the launch behind `roofline` scanned (rank 0)
    if world > 1:
        # N > 1: the headline is configs[2] - the same N reads split evenly; the roofline kernel is rank 0's candidate launch on its share
        value = strong["value"]
        head_reads = par.shard_range(N, 0, world, align=1)[1]
        n_launch = max(strong_kms[1], 1)
        strands_per_launch = 2.0 * args.steps / n_launch
        cand_ms = strong_kms[0] / n_launch
    pass_ms = (kms["count"][0] + kms["offsets"][0] + kms["fill"][0]) / (2.0 * args.steps)    # one strand's share of a step's kernels
    alg_bytes = N * L + hits_per_pass * 14 + K * 8
    pass_gbs = alg_bytes / (pass_ms * 1e-3) / 1e9
    cand_flops = 2.0 * 4 * PL * K * float(head_reads) * Lout * strands_per_launch
    cand_tflops = cand_flops / (cand_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    # the PMC entry of the launch that was timed: same kernel, same grid (blocks of 8 reads, tools/summarize_traffic.py keys by shape)
    cand_blocks = (N + 7) // 8                                  # (the PMC driver tools/prof_scan.py scans one strand per call)
    for tf in TRAFFIC_FILES:
        if traffic is None and os.path.exists(tf) and (N, L, K, PL) == (100_000, 200, 200, 12):
            with open(tf) as fh:
                tj = json.load(fh)
            for e in tj.get("launches", []):
                if e["kernel"].startswith("scan_cand_kernel") and e["blocks"] == cand_blocks:
                    traffic = e["hbm_bytes_per_launch"]
                    traffic_src = {"file": os.path.relpath(tf, ROOT), "commit": tj.get("commit"), "command": tj.get("command"),
                                   "kernel": e["kernel"], "blocks": e["blocks"], "launches_averaged": e["launches"]}
    out = {
        "metric": "bases scanned/sec",
        "value": value,
        "unit": "bases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": strong["ms_per_step"] if world > 1 else dt / args.steps * 1e3,
        "ms_per_step_default_mode": dt_default / args.steps * 1e3,
        "ms_per_step_default_mode_note": "the same shard and step count with motifs_ctx_set_records_in_stream_order off (the library default: every call returns "
                                         "with its records written); `ms_per_step` / `value` are measured with it on (config.records_in_stream_order); at N > 1 this "
                                         "is the weak-scaling shard, not the configs[2] split",
        "cold_start_ms_per_step": sum(cold) / len(cold),
        "cold_start_ms_each": cold,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": (f"PWM log-odds scan, both strands, {N} seqs x {L} bp per GPU vs {K} PWMs len {PL} "
                         "(BASELINE configs[1]); hits thresholded (>0) and compacted on device in reference order") if world == 1 else
                        (f"PWM log-odds scan, both strands, {N} seqs x {L} bp IN TOTAL split evenly over {world} GPUs vs {K} PWMs len {PL} "
                         "(BASELINE configs[2]: sharded reads + RCCL sum of the hit histogram); hits thresholded (>0) and compacted on device, "
                         "records in the reference's per-(PWM, read) order"),
            "seqs_per_gpu": N if world == 1 else head_reads, "seqs_total": N * (1 if world > 1 else world), "seq_len": L, "pwms": K, "pwm_len": PL,
            "hits_per_step": int(tot_hits.item()) if world == 1 else int(strong.get("hist_total_hits", 0)),
            "parallelism": f"sequence shards x{world}; per step one sum of the 2 x {K} hit histogram: {reducer.kind} ({reducer_note})",
            "untimed_preheat_steps": PREHEAT,
            "records_in_stream_order": True,
        },
        "roofline": {
            "kernel": "scan_cand_kernel_q<3,4,2,compact> (v_mfma_f32_32x32x16_f16 candidate filter, four reads per wave; "
                      f"{strands_per_launch:.0f} strand(s) of the shard per launch)",
            "strands_per_launch": strands_per_launch,
            "bound": "mfma",
            "achieved": cand_tflops,
            "peak": MFMA_F16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": cand_tflops / MFMA_F16_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "flops_per_launch": cand_flops,
            "avg_launch_ms": cand_ms,
            "note": "per tile of a wave: 12 matrix instructions (384 cycles of the matrix pipe) and 86 vector instructions (one v_alignbit per sign "
                    "bit) on the same issue port, ISA count in DESIGN 2.7-2 - the two are level, which caps this fraction near 0.6",
        },
        "pass_hbm": {
            "kernels": "scan_cand_kernel + stage_hits + row scan + emit_records (one strand pass)",
            "bound": "hbm", "achieved": pass_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pass_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes": alg_bytes, "avg_pass_ms": pass_ms,
            "note": "fused-hits contract (N*L codes in, 14 B per hit out, no dense score tensor): the pass is bound "
                    "by the matrix cores and the exact re-scoring, not by HBM; dense_kernel is the a17 dense-score contract",
        },
        "strong": strong,
        "weak": weak_line,
        "kernel_ms_per_step": {"scan_cand": kms["count"][0] / args.steps, "stage_hits_row_scan": kms["offsets"][0] / args.steps,
                               "emit_records": kms["fill"][0] / args.steps},
    }
    out.update(extras)
    if train is not None:
        out["train"] = train

    if not args.no_cpu and world == 1:
        from oracle import scan_oracle as so

        # (1) the optimised port: AVX2/F16C, 8 PWMs per register, same rounding sequence, OpenMP on every host core;
        #     one warm-up, best of three
        ns = min(args.cpu_sample, N)
        best, c_hits = None, 0
        for rep in range(4):
            t0 = time.perf_counter()
            got = [so.get_pos_scores_arr_fast(bank, lens, codes[:ns], rc=rc, cap_hint=int(max(need) * ns / N * 1.2) + 1024) for rc in (False, True)]
            t = time.perf_counter() - t0
            if got[0] is None:
                break
            c_hits = sum(len(g[0]) for g in got)
            if rep > 0:
                best = t if best is None else min(best, t)
        # (2) the literal restatement (dense (K, nb, 4L) tensor, soft binary16 per multiply and add, findall), one run
        nl = min(args.cpu_literal_sample, N)
        onehot = sy.codes_to_onehot(codes[:nl])
        t0 = time.perf_counter()
        lit = [so.get_pos_scores_arr(bank, lens, onehot, rc=rc) for rc in (False, True)]
        ldt = time.perf_counter() - t0
        # both against the GPU records of the same reads
        g_ns = sum(int((hits[rc][: need[rc], 1] <= ns).sum().item()) for rc in (0, 1))
        g_nl = sum(int((hits[rc][: need[rc], 1] <= nl).sum().item()) for rc in (0, 1))
        assert sum(len(x[0]) for x in lit) == g_nl, "literal CPU port and GPU disagree on the sample"
        if best is not None:
            assert c_hits == g_ns, "vectorised CPU port and GPU disagree on the sample"
            out["cpu_baseline"] = {
                "value": ns * L / best, "unit": "bases/s", "cores": so.num_threads(), "physical_cores": so.physical_cores(), "kind": "port",
                "sample": f"first {ns} sequences x {L} bp, both strands; AVX2/F16C port of the reference arithmetic (8 PWMs per register, 5-8 independent chains in flight, "
                          f"binary16 rounding after every add, records in findall order), OpenMP, 1 warm-up + best of 3: {best:.2f} s; "
                          f"hits {c_hits} == GPU on the same reads",
                "literal": {"value": nl * L / ldt, "unit": "bases/s", "kind": "port-literal", "cores": so.num_threads(),
                            "sample": f"first {nl} sequences, dense (K, nb, 4L) soft-binary16 tensor + findall as the reference does it, one run: {ldt:.2f} s"},
            }
        else:
            out["cpu_baseline"] = {"value": nl * L / ldt, "unit": "bases/s", "cores": so.num_threads(), "kind": "port",
                                   "sample": f"host CPU lacks AVX2/F16C: literal soft-binary16 port on the first {nl} sequences, {ldt:.2f} s"}
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if train is not None:
            # the reference-faithful float64 restatement (full-lag conv_code_diff etc.) needs ~50 minutes per mini-batch at
            # configs[1], so the CPU leg of the train step is timed at configs[0] shape (100 bp, 32 filters of length 8)
            # for BOTH sides
            from oracle import model_oracle as mo
            import torch as _t

            hp1 = mo.Hyperparam(filter_len=8, M=32)
            rng = np.random.default_rng(1)
            c1 = rng.integers(0, 4, size=(6, 100)).astype(np.uint8)
            cdl1 = mo.UCDL(hp1, rng)
            t0 = time.perf_counter()
            mo.loss_and_grads(c1, cdl1, hp1, _t.float32)
            ct = time.perf_counter() - t0
            g1 = pkg.model.ucdl(pkg.model.Hyperparam(filter_len=8, M=32), 100, ctx=ctx, seed=1, arena_bytes=8 << 30)
            cc = sy.gen_codes(64 * 6, 100, 3)
            g1.model.train_step(cc, 64, want_l1=False)
            t0 = time.perf_counter()
            g1.model.train_step(cc, 64, want_l1=False)
            gt = time.perf_counter() - t0
            out["train"]["cpu_baseline_cfg0"] = {
                "value": 6 / ct, "unit": "seqs/s", "cores": _t.get_num_threads(), "kind": "port",
                "sample": f"one mini-batch fwd+bwd of the torch-CPU restatement at configs[0] shape, {ct:.1f} s",
                "gpu_same_shape_seqs_per_s": 64 * 6 / gt,
            }
            g1.model.close()
            # ... and beside the configs[1] train step itself: the oracle's needed-lag / direct-syntax forms (the same sums as the
            # literal graph, tests/test_oracle_model.py::test_needed_lag_update_D_equals_the_literal_one; the literal forms need
            # ~4 minutes per mini-batch forward at this shape) in float32 on torch's CPU threads, forward + backward
            mo.NEEDED_LAGS = mo.FAST_SYNTAX = True
            try:
                hp2 = mo.Hyperparam(filter_len=args.filter_len, M=args.filters)
                cdl2 = mo.UCDL(hp2, np.random.default_rng(2))
                c2 = np.random.default_rng(3).integers(0, 4, size=(6, L)).astype(np.uint8)
                mo.loss_and_grads(c2, cdl2, hp2, _t.float32)
                t0 = time.perf_counter()
                n_mb = 0
                while n_mb < 3 or (time.perf_counter() - t0 < 10 and n_mb < 12):
                    mo.loss_and_grads(c2, cdl2, hp2, _t.float32)
                    n_mb += 1
                ct2 = (time.perf_counter() - t0) / n_mb
            finally:
                mo.NEEDED_LAGS = mo.FAST_SYNTAX = False
            out["train"]["cpu_baseline"] = {
                "value": 6 / ct2, "unit": "seqs/s", "cores": _t.get_num_threads(), "kind": "port",
                "sample": f"{n_mb} mini-batches of 6 reads x {L} bp, {args.filters} filters of length {args.filter_len}: forward + backward of the torch-CPU "
                          f"restatement in its needed-lag / direct-syntax forms, float32, {ct2:.2f} s per mini-batch",
                "gpu_over_cpu": out["train"]["seqs_per_s"] / (6 / ct2),
            }
    # the driver keeps the tail of the line: the figures a reader looks for first go last
    if train is not None:
        tr = out.pop("train")
        out["train"] = tr
        out["train_ms_per_step"] = tr["ms_per_step"]
        out["train_ms_per_step_g1"] = tr["ms_per_step_g1"]
        out["code_retrieval_reads_per_s"] = tr["code_retrieval"]["reads_per_s"]
    if "dense_kernel" in out:
        out["dense_frac"] = out["dense_kernel"]["frac"]
    out["cfg3_shard_ms"] = out.get("cfg3_shard", {}).get("ms_per_step_both_strands")
    out["cfg4_shard_ms"] = out.get("cfg4_shard", {}).get("ms_per_step_both_strands")
    out["ms_per_step_again"] = out["ms_per_step"]
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
