#!/usr/bin/env python3
"""bench.py -- Viterbi Mframes/s at S=361, T=30000 on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic songs per GPU:
forward max-plus recursion + back-trace for [B=128, T=30000, S=361] float32 log-emissions
already resident in HBM, plus (N > 1) the RCCL gather of the decoded paths on rank 0.
Per-GPU work is fixed as N grows (weak scaling: 128 songs per GPU = BASELINE configs[2],
and configs[3] = 1024 songs over 8 GPUs).

The forward pass and the back-trace of a step run back to back on one HIP stream (overlapping the back-trace of
step i with the forward pass of step i+1 on a second stream was measured: the HBM-bound back-trace slows the
latency-bound forward pass from 10.6 to 13.7 ms, a net loss).  Only the gather (N > 1) is overlapped: it is
launched non-blocking on the communicator's stream after the back-trace of step i and completes under the
forward pass of step i+1; every batch in flight has its own workspace, path buffer and gather buffer (two
slots).  The timed region ends with a device synchronisation and the completion of every gather.
`--serial` waits for each gather before the next step starts.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The CPU baseline leg (rank 0, N == 1 only) runs the oracle --
the NumPy restatement of the reference's loop, one thread -- on a bounded sample of the same
songs and doubles as a parity check of the GPU result.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from viterbi_spl_amd import ViterbiDecoder, sharded, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="songs per GPU")
    ap.add_argument("--frames", type=int, default=30000)
    ap.add_argument("--states", type=int, default=361)
    ap.add_argument("--algo", default="auto", choices=["auto", "dense", "banded"])
    ap.add_argument("--emissions", default="peaks", choices=["peaks", "dense"])
    ap.add_argument("--transition", default="tonet", choices=["tonet", "dense", "durrieu"])
    ap.add_argument("--f16", action="store_true", help="store emissions as float16")
    ap.add_argument("--dmax", type=int, default=14, help="band half-width of the tonet-recipe transition (tonet 14, jdc 40, imm 56)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--serial", action="store_true", help="no overlap between consecutive steps (one stream)")
    return ap.parse_args()


def make_params(args):
    S = args.states
    if args.transition == "tonet":
        return synth.log_params(synth.tonet_transition(S - 1, args.dmax), synth.floored_prior(S))
    if args.transition == "durrieu":   # imm's own decoder: dense, piecewise constant in 20-bin distance bands, uniform prior
        A = synth.durrieu_transition(S - 1, 20)
        return (np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(S, 1.0 / S)).astype(np.float32))
    return synth.dense_random_log_transition(S, seed=3), synth.dense_random_log_transition(S, seed=4)[0].copy()


def cpu_baseline(logA_T, log_pi, E, gpu_states, gpu_loglik, seconds):
    """Oracle (NumPy restatement of imm/tf_viterbi.py:91-107, one thread) on the first songs of the
    batch until ~`seconds` of CPU time; also the C restatement on all cores.  Checks parity."""
    from oracle import viterbi_oracle as vo
    T = E.shape[1]
    done, frames, t_used, exact = 0, 0, 0.0, True
    while done < E.shape[0] and t_used < seconds:
        e = E[done].float().cpu().numpy()
        t0 = time.perf_counter()
        st, ll = vo.decode_numpy(logA_T, log_pi, e)
        t_used += time.perf_counter() - t0
        exact = exact and bool(np.array_equal(st, gpu_states[done].cpu().numpy())) and bool(np.float32(ll) == np.float32(gpu_loglik[done].item()))
        frames += T
        done += 1
    out = {"value": frames / t_used / 1e6, "unit": "Mframes/s", "cores": 1, "kind": "port",
           "sample": f"first {done} of the batch's songs, T={T}, NumPy float32 loop (oracle/viterbi_oracle.py::decode_numpy)",
           "bit_exact_vs_gpu": exact, "numpy": np.__version__}
    nthr = min(vo.num_threads(), 16)
    nb = min(E.shape[0], nthr)
    e = E[:nb].float().cpu().numpy()
    t0 = time.perf_counter()
    st, ll = vo.decode_c(logA_T, log_pi, e, threads=nthr)
    dt = time.perf_counter() - t0
    exact_c = bool(np.array_equal(st, gpu_states[:nb].cpu().numpy())) and bool(np.array_equal(ll, gpu_loglik[:nb].cpu().numpy()))
    out_c = {"value": nb * T / dt / 1e6, "unit": "Mframes/s", "cores": nthr, "kind": "port",
             "sample": f"{nb} songs, T={T}, scalar C restatement, one song per thread (oracle/viterbi_oracle.c)",
             "bit_exact_vs_gpu": exact_c}
    return out, out_c


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    B, T, S = args.batch, args.frames, args.states
    logA_T, log_pi = make_params(args)
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    algo = args.algo
    if algo == "auto":
        algo = "banded" if dec.info["banded_ok"] else "auto"      # "auto": the step-structured kernel if the plan proves it, else dense
    gen = synth.emissions_peaks if args.emissions == "peaks" else synth.emissions_dense
    dt = torch.float16 if args.f16 else torch.float32
    E = gen(B, T, S, seed=1234, device=dev, dtype=dt, first_song=rank * B)
    n_total = B * world
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"     # the latter: rehearse the gather path on one GPU
    NSLOT = 2 if use_dist else 1                   # batches in flight (the gather of step i overlaps step i+1)
    for k in range(NSLOT):
        dec._workspace(B, T, k)                     # allocate before anything is timed
    states_k = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(NSLOT)]
    loglik_k = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(NSLOT)]
    if use_dist and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    if use_dist and rank == 0:
        out_k = [(torch.empty((world, B, T), dtype=torch.int32, device=dev), torch.empty((world, B), dtype=torch.float32, device=dev))
                 for _ in range(NSLOT)]
    else:
        out_k = [(None, None)] * NSLOT
    pending = [None] * NSLOT
    gather_mode = ["async"]

    def step(i, ev=None, overlap_gather=True):
        """forward(i), back-trace(i) on the current stream; the gather of the paths on the communicator's stream."""
        k = i % NSLOT
        if pending[k] is not None:                     # gather(i - NSLOT) still reads states_k[k]
            for w in pending[k]:
                w.wait()
            pending[k] = None
        if ev is not None:
            ev[0].record()
        dec.decode_into(E, states_k[k], loglik_k[k], algo=algo, phase="forward", slot=k)
        if ev is not None:
            ev[1].record()
        dec.decode_into(E, states_k[k], loglik_k[k], algo=algo, phase="backtrace", slot=k)
        if ev is not None:
            ev[2].record()
        if use_dist:
            if gather_mode[0] == "async":
                try:
                    pending[k] = sharded.gather_paths_async(states_k[k], loglik_k[k], out_k[k][0], out_k[k][1], dst=0)
                except Exception as exc:           # defensive: fall back to the plain blocking gather
                    if rank == 0:
                        print(f"bench: non-blocking gather unavailable ({exc!r}); using the blocking gather", file=sys.stderr)
                    gather_mode[0] = "blocking"
            if gather_mode[0] == "blocking":
                sharded.gather_paths(states_k[k], loglik_k[k], n_total, dst=0)
            elif not overlap_gather:
                for w in pending[k]:
                    w.wait()
                pending[k] = None

    def step_serial(i, ev=None):
        step(i, ev, overlap_gather=False)

    def drain():
        for k in range(NSLOT):
            if pending[k] is not None:
                for w in pending[k]:
                    w.wait()
                pending[k] = None
        torch.cuda.synchronize()

    def timed(step_fn, n, with_events):
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)] if with_events else [None] * n
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step_fn(i, evs[i])
        drain()
        if use_dist:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, evs

    step_fn = step_serial if args.serial else step
    for k in range(NSLOT):             # prime every slot once (code objects, first touch of the workspaces) whatever --warmup says
        step_fn(k)
    drain()
    for i in range(args.warmup):
        step_fn(i)
    drain()
    elapsed, events = timed(step_fn, args.steps, True)
    states, loglik = states_k[(args.steps - 1) % NSLOT], loglik_k[(args.steps - 1) % NSLOT]

    fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    bt_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))

    if rank == 0:
        esize = 2 if args.f16 else 4
        frames_per_launch = B * T
        fwd_bytes = frames_per_launch * (S * esize + S * 2)          # emission row in, uint16 back-pointer row out
        bt_bytes = frames_per_launch * (2 + 4)                       # one back-pointer in, one int32 state out
        value = n_total * T * args.steps / elapsed / 1e6
        achieved = fwd_bytes / (fwd_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{algo}_B{B}_T{T}_S{S}_{'f16' if args.f16 else 'f32'}"
        if os.path.exists(tf):       # PMC measurement of the same command, collected by scripts/pmc_traffic.sh
            try:
                traffic = json.load(open(tf)).get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        bt_traffic = None
        if os.path.exists(tf):
            try:
                bt_traffic = json.load(open(tf)).get(key, {}).get("backtrace", {}).get("hbm_bytes_per_launch")
            except Exception:
                bt_traffic = None
        SD = (S + 5) // 4 * 4
        fwd_kernel = "step4_forward_kernel" if (dec.info["step_ok"] and args.algo == "auto") else "dense_forward_kernel"
        if algo == "banded":   # same rule as launch_banded_t (kernels.hip)
            nwt = next((w for w in (2, 4, 6, 8, 12) if w * 64 >= S), 0)
            floor_form = dec.info["floor_ok"] and dec.info["n_dense_rows"] == 0 and S < nwt * 64
            fwd_kernel = "banded_forward_kernel"
            if floor_form:
                fwd_kernel = "banded_floor_pair_forward_kernel" if (dec.info["pair_ok"] and B > 256 and nwt <= 8 and dec.info["group_window"] <= 32) else "banded_floor_forward_kernel"
        out = {
            "metric": "Viterbi Mframes/s at S=361 T=30k; achieved HBM GB/s vs peak",
            "value": value, "unit": "Mframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch of {B} songs per GPU, T={T}, S={S}, {'fp16' if args.f16 else 'fp32'} log-emissions "
                                   f"(BASELINE configs[2]; configs[3] at 8 GPUs), forward + back-trace"
                                   + (" + RCCL gather of paths" if use_dist else ""),
                       "songs_per_gpu": B, "frames": T, "states": S, "emissions": args.emissions,
                       "transition": args.transition, "band_half_width": args.dmax if args.transition == "tonet" else None,
                       "forward_kernel": algo, "plan": dec.info},
            "roofline": {"bound": "hbm", "kernel": fwd_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": fwd_bytes, "avg_launch_ms": fwd_ms,
                         "bytes_per_frame": S * esize + S * 2,
                         "note": "algorithmic bytes per SURVEY 8d (emission row in + uint16 back-pointer row out); this "
                                 "implementation stores the float32 delta row instead (lazy back-pointers, DESIGN.md), "
                                 "and the recursion is fp32-VALU / latency bound, not HBM bound"},
            "kernels_ms": {"forward": fwd_ms, "backtrace": bt_ms},
            "gather": (None if not use_dist else ("blocking" if (args.serial or gather_mode[0] == "blocking") else
                       "non-blocking on the communicator's stream, completes under the next step's forward pass; two path-buffer slots")),
            "backtrace": {"algorithmic_bytes_per_launch": bt_bytes, "traffic": bt_traffic,
                          "implementation_bytes_per_frame": SD * 4,
                          "hbm_gbs_from_traffic": (bt_traffic / (bt_ms * 1e-3) / 1e9) if bt_traffic else None},
            "whole_path_bytes_per_frame": S * esize + S * 2 + 6,
            "whole_path_hbm_frac": value * 1e6 * (S * esize + S * 2 + 6) / 1e9 / (HBM_PEAK_GBS * world),
        }
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.synchronize()
            cb, cbc = cpu_baseline(logA_T, log_pi, E, states, loglik, args.cpu_seconds)
            out["cpu_baseline"] = cb
            out["cpu_baseline_c"] = cbc
            out["gpu_over_cpu_1core"] = value / cb["value"]
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
