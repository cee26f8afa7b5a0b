#!/usr/bin/env python3
"""bench.py -- Viterbi Mframes/s at S=361, T=30000 on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic songs per GPU: forward max-plus recursion +
back-trace for [B=128, T=30000, S=361] float32 log-emissions already resident in HBM, plus (N > 1) the RCCL gather
of the decoded paths on rank 0.  Per-GPU work is fixed as N grows (weak scaling: 128 songs per GPU = BASELINE
configs[2]; configs[3] = 1024 songs over 8 GPUs).

Schedule of the timed region: the forward pass of step i runs on one HIP stream, its back-trace on a second one
(ordered by an event), so the back-trace of step i -- 0.6 ms on the half of the chip the 128 one-song workgroups
leave idle -- overlaps the forward pass of step i+1; every batch in flight has its own workspace, path buffer and
gather buffer (two slots).  With N > 1 the gather of step i is launched non-blocking on the communicator's stream
behind its back-trace.  The region ends with a device synchronisation and the completion of every gather: all K
steps are complete inside it.  `--serial` runs forward, back-trace and gather of a step back to back on one stream.

Rank 0 prints ONE JSON line (always the last line of stdout).  At N = 1 it also carries, measured in the same process:
  "sweep"    B = 256 / 512 / 1024 / 2048 at S=361 fp32 (the saturation regime; songs repeat with period 32): serial and
             overlapped schedules, the wave form with its half and its full delta history, dense emissions, a ragged batch,
             and the back-trace's data-dependent event counts per 1000 frames,
  "configs4" [256, 30000, 722] fp16 emissions (BASELINE configs[4]): jdc band (d_max 40) and the Durrieu matrix,
  "cpu_baseline" the oracle (NumPy restatement of the reference's loop, one thread) on a bounded sample of the same
             songs -- which doubles as a parity check of the GPU result -- and the C restatement on all cores.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun's environment: launches its own N workers,
                                                            launch_workers() below, and relays rank 0's JSON line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
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


def launch_workers(n, argv, child=None, timeout=None):
    """Start `n` worker processes of `child` (default: this file) under torch.distributed.run on this node (one rank per GPU,
    rendezvous on 127.0.0.1 at a free port, HSA_ENABLE_IPC_MODE_LEGACY=0 kept for RCCL), pass their output through, and print
    the LAST JSON line of their stdout (rank 0's result) as this process's own last stdout line.  Returns the workers' exit
    code (non-zero if any rank failed, or if no JSON line came back).  The calling process must not have initialised the
    GPU: the workers are children, nothing is re-executed in place."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), child or os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last_json = None
    try:
        for line in proc.stdout:
            ls = line.strip()
            if ls.startswith("{") and ls.endswith("}"):
                if last_json is not None:
                    print(last_json, flush=True)       # an earlier JSON-looking line is ordinary output after all
                last_json = ls
            else:
                sys.stdout.write(line)
                sys.stdout.flush()
        rc = proc.wait(timeout=timeout)
    except BaseException:
        proc.kill()
        proc.wait()
        raise
    if last_json is not None:
        print(last_json, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="songs per GPU")
    ap.add_argument("--frames", type=int, default=30000)
    ap.add_argument("--states", type=int, default=361)
    ap.add_argument("--algo", default="auto", choices=["auto", "dense", "banded", "wave", "group"])
    ap.add_argument("--emissions", default="peaks", choices=["peaks", "dense"])
    ap.add_argument("--transition", default="tonet", choices=["tonet", "dense", "durrieu"])
    ap.add_argument("--f16", action="store_true", help="store emissions as float16")
    ap.add_argument("--dmax", type=int, default=14, help="band half-width of the tonet-recipe transition (tonet 14, jdc 40, imm 56)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sweep / configs4 blocks")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes (roofline.traffic from profiles/traffic.json)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--serial", action="store_true", help="one stream: no overlap between consecutive steps")
    ap.add_argument("--option", action="append", default=[], help="vit_plan_set_option key=value (kernel-selection override)")
    return ap.parse_args()


def make_params(transition, S, dmax):
    if transition == "tonet":
        return synth.log_params(synth.tonet_transition(S - 1, dmax), synth.floored_prior(S))
    if transition == "durrieu":   # imm's own decoder: dense, piecewise constant in 20-bin distance bands, uniform prior
        A = synth.durrieu_transition(S - 1, 20)
        return (np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(S, 1.0 / S)).astype(np.float32))
    return synth.dense_random_log_transition(S, seed=3), synth.dense_random_log_transition(S, seed=4)[0].copy()


def bound_of(kernel):
    """roofline.bound: what binds the forward kernel ("hbm" only where it is HBM; `frac` is quoted against the HBM roofline whatever
    this says, because that is the metric)."""
    if kernel == "wave_forward_kernel":
        return "hbm"          # full history: 5.2-5.5 TB/s of real traffic; (half history: vector issue -- the sweep rows say which)
    return "latency" if kernel.startswith("banded") else "issue"


def limited_by(kernel, B):
    """What actually binds the forward kernel (the roofline fraction is quoted against HBM whatever this says)."""
    if kernel == "wave_forward_kernel":
        return "vector-instruction issue (one song per wavefront: 253 VALU instructions per frame; HBM traffic next)"
    if kernel.startswith("banded"):
        return ("latency: one barrier-synchronised frame per ~810 cycles of a song's workgroup (dependent chain barrier -> LDS reads -> "
                "VALU -> LDS writes -> barrier)" + ("; 128 songs occupy 128 of the 256 CUs" if B <= 128 else ""))
    return "vector / LDS instruction issue"


def forward_kernel_name(dec, algo, B, S):
    """Name of the forward kernel the library launches for (plan, algo, batch): same rules as capi.hip / kernels.hip."""
    info = dec.info
    if algo in ("auto", "banded", "wave", "group") and info["banded_ok"]:
        nwt = next((w for w in (2, 4, 6, 8, 12) if w * 64 >= S), 0)
        if dec.forward_family(B, algo) == "wave":      # (vit_forward_family: the batch-size threshold scales with the device's compute units)
            return "wave_forward_kernel"
        floor_form = info["floor_ok"] and info["n_dense_rows"] == 0 and S < nwt * 64
        if not floor_form:
            return "banded_forward_kernel"
        pair = info["pair_ok"] and B > 256 and nwt <= 8 and info["group_window"] <= 32
        return "banded_floor_pair_forward_kernel" if pair else "banded_floor_forward_kernel"
    if algo == "auto" and info["step_ok"]:
        return "step4s_forward_kernel"
    return "dense_forward_kernel"


def time_serial(dec, E, algo, steps, warmup=1, lengths=None):
    """forward + back-trace of one batch back to back on the current stream; HIP events around each half."""
    B, T, _ = E.shape
    st = torch.empty((B, T), dtype=torch.int32, device=E.device)
    ll = torch.empty((B,), dtype=torch.float32, device=E.device)
    for _ in range(warmup):
        dec.decode_into(E, st, ll, lengths, algo=algo)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        dec.decode_into(E, st, ll, lengths, algo=algo, phase="forward")
        ev[i][1].record()
        dec.decode_into(E, st, ll, lengths, algo=algo, phase="backtrace")
        ev[i][2].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    fwd = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    bt = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    return {"wall_ms_per_step": wall / steps * 1e3, "forward_ms": fwd, "backtrace_ms": bt}, st, ll


# configs[4] kernels, per frame and song (one song per workgroup), from the SQ counters of profiles/r04_pmc_configs4.txt
# ([256, 30000, 722] fp16, ten launches): SQ_INSTS_VALU / frames, SQ_LDS_IDX_ACTIVE / frames
C4_ISSUE_PROFILE = {
    "jdc_band_dmax40": {"kernel": "banded_floor_forward_kernel<84,12,1,4,half>", "valu_instructions_per_frame_and_song": 1302,
                        "lds_active_cycles_per_frame_and_song": 1329, "source": "profiles/r04_pmc_configs4.txt"},
    "durrieu_dense": {"kernel": "step4s_forward_kernel<20,9,2,half>", "valu_instructions_per_frame_and_song": 1020,
                      "lds_active_cycles_per_frame_and_song": 1268, "source": "profiles/r04_pmc_configs4.txt"},
}
VALU_LANE_RATE = 256 * 4 * 16 * 2.4e9   # lane-instructions per second: 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz


def valu_ceiling(dec, S, frames, fwd_ms):
    """Frames per second if the forward kernel were bound by vector-instruction issue alone: candidates per state (window + floor
    + extra columns), one lane-instruction each (half a v_pk_add_f32 + half a v_max3_f32), nothing else counted."""
    info = dec.info
    if not info.get("banded_ok"):
        return None
    cand = info["max_window"] + 1 + info["n_extras"]
    per_frame = S * cand
    peak = VALU_LANE_RATE / per_frame
    return {"candidates_per_state": cand, "lane_instructions_per_frame": per_frame, "peak_Mframes_per_s": peak / 1e6,
            "achieved_Mframes_per_s": frames / (fwd_ms * 1e-3) / 1e6, "frac": frames / (fwd_ms * 1e-3) / peak}


def time_overlapped(dec, E, algo, steps, warmup=1, lengths=None):
    """The headline's schedule on any batch: the back-trace of step i on a second stream under the forward pass of step i+1
    (two workspace slots, two path buffers); wall time per step over `steps` complete steps."""
    B, T, _ = E.shape
    dev = E.device
    st = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(2)]
    ll = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(2)]
    for k in range(2):
        dec._workspace(B, T, k, algo)
    s_fwd, s_bt = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)   # forward workgroups are dispatched first
    bt_done = [None, None]
    dec.set_option("bt_chunks", dec.chunks_beside_forward(B))      # the back-trace on the units the forward pass leaves idle
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    small_groups = dec.history_mode(B, T, algo) == "half" and B <= 4 * n_cus
    if small_groups:            # one forward wave per SIMD leaves 256 registers free: eight-wave back-trace workgroups (208) can start beside them,
        dec.set_option("bt_block_waves", 8)     # sixteen-wave ones (416) cannot (B = 1024, half history: 22.1 -> 20.2 ms per step)

    def run(n):
        for i in range(n):
            k = i & 1
            with torch.cuda.stream(s_fwd):
                if bt_done[k] is not None:
                    s_fwd.wait_event(bt_done[k])
                dec.decode_into(E, st[k], ll[k], lengths, algo=algo, phase="forward", slot=k)
                fwd_done = torch.cuda.Event()
                fwd_done.record()
            with torch.cuda.stream(s_bt):
                s_bt.wait_event(fwd_done)
                dec.decode_into(E, st[k], ll[k], lengths, algo=algo, phase="backtrace", slot=k)
                bt_done[k] = torch.cuda.Event()
                bt_done[k].record()
        torch.cuda.synchronize()

    run(warmup)
    t0 = time.perf_counter()
    run(steps)
    wall = (time.perf_counter() - t0) / steps * 1e3
    dec._ws_slots.clear()
    dec.set_option("bt_chunks", 0)
    if small_groups:
        dec.set_option("bt_block_waves", 0)
    return wall, st[(steps - 1) & 1], ll[(steps - 1) & 1]


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
           "bit_exact_vs_gpu": exact, "numpy": np.__version__, "host": host_info()}
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    nthr = max(1, min(vo.num_threads(), usable, E.shape[0]))      # every usable core, one song per thread (never more threads than songs)
    nb = min(E.shape[0], nthr)
    e = E[:nb].float().cpu().numpy()
    t0 = time.perf_counter()
    st, ll = vo.decode_c(logA_T, log_pi, e, threads=nthr)
    dt = time.perf_counter() - t0
    exact_c = bool(np.array_equal(st, gpu_states[:nb].cpu().numpy())) and bool(np.array_equal(ll, gpu_loglik[:nb].cpu().numpy()))
    out_c = {"value": nb * T / dt / 1e6, "unit": "Mframes/s", "cores": nthr, "kind": "port",
             "sample": f"{nb} songs, T={T}, scalar C restatement, one song per thread (oracle/viterbi_oracle.c)",
             "usable_cores": usable, "bit_exact_vs_gpu": exact_c}
    return out, out_c


def host_info():
    """nproc and CPU model of the box the CPU baseline runs on (the reference's own benchmark,
    dcnet/tf_viterbi_decoding.py:266-282, is a host-CPU number)."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    return {"nproc": os.cpu_count(), "usable_cores": usable, "cpu_model": model}


def measure_traffic(B, T, algo, options):
    """HBM bytes per launch of the forward and back-trace kernels from the PMC counters, collected in this run: two
    rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE: they do not fit one pass, and no tracing domain is combined
    with them) over a child process that decodes the same [B, T, 361] tonet batch twice (scripts/prof_target.py; songs
    repeat with period 32).  gfx950: FETCH_SIZE counts 64 B per 128-byte request of a wide coalesced stream, so read
    bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 is exact.  Returns None if rocprofv3 is not
    available or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="vit_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "--",
                   sys.executable, os.path.join(ROOT, "scripts", "prof_target.py"), str(B), algo, "2", str(T)] + list(options)
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=150)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(tmp, counter, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    k = row["Kernel_Name"]
                    if "vit::" in k and row["Counter_Name"] == counter:
                        name = k.split("vit::")[1].split("<")[0].split("(")[0]
                        sums.setdefault((name, counter), []).append(float(row["Counter_Value"]))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    launches = 2
    out = {}
    for (name, counter), vals in sums.items():
        out.setdefault(name, {})[counter + "_KiB_per_launch"] = sum(vals) / launches      # (the verify pass is its own dispatch: summed)
    for name, d in out.items():
        d["hbm_bytes_per_launch"] = int(2 * d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024 + d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024)
    return out or None


def tiled_emissions(gen, B, T, S, seed, dev, dtype, period=32):
    """[B,T,S] emissions whose songs repeat with `period` (generating 2048 distinct 30000-frame songs would take longer
    than the measurement; identical songs at different addresses cost the kernels the same)."""
    base = gen(min(B, period), T, S, seed=seed, device=dev, dtype=dtype)
    if B <= period:
        return base
    return base.repeat((B + period - 1) // period, 1, 1)[:B].contiguous()


def oracle_spot_check(logA_T, log_pi, E, states, loglik, songs):
    from oracle import viterbi_oracle as vo
    rs, rl = vo.decode_c(logA_T, log_pi, E[songs].float().cpu().numpy())
    return bool(np.array_equal(states[songs].cpu().numpy(), rs)) and bool(np.array_equal(loglik[songs].cpu().numpy(), rl))


def per_1000_frames(dec, B, T, frames=None, slot=0):
    ct = dec.backtrace_counters(B, T, slot)
    n = float(frames if frames is not None else B * T)
    return {k: round(v * 1000.0 / n, 3) for k, v in ct.items()}


def sweep_row(dec, logA_T, log_pi, E, algo, steps, lengths=None, overlapped=True, options=None):
    """One row of the sweep: forward + back-trace of `E` back to back on one stream (HIP events around each half, `steps`
    timed steps) and, where two workspace slots fit, the two-stream schedule; fractions of the HBM roofline on ALGORITHMIC
    bytes; the back-trace's event counts; an oracle spot check of three songs."""
    B, T, S = E.shape
    esz = E.element_size()
    dec.set_option("reset", 0)
    for k, v in (options or {}).items():
        dec.set_option(k, v)
    frames = int(lengths.sum().item()) if lengths is not None else B * T
    r, st, ll = time_serial(dec, E, algo, steps=steps, lengths=lengths)
    ser_ms = r["forward_ms"] + r["backtrace_ms"]
    r.update({"songs": B, "frames_decoded": frames, "Mframes_per_s": frames / ser_ms / 1e3,
              "forward_kernel": forward_kernel_name(dec, algo, B, S),
              "history": dec.history_mode(B, T, algo),
              "workspace_GB": dec.workspace_bytes(B, T, algo) / 1e9,
              "forward_hbm_frac": frames * (S * esz + S * 2) / (r["forward_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "whole_path_hbm_frac": frames * (S * esz + S * 2 + 6) / (ser_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "backtrace_events_per_1000_frames": per_1000_frames(dec, B, T, frames)})
    vc = valu_ceiling(dec, S, frames, r["forward_ms"])
    if vc:
        r["forward_valu_frac"] = vc["frac"]
    if overlapped and lengths is None and S == 361 and esz == 4:      # (the peak-sparse main rows) HBM bytes of the same kernels and shapes from the committed PMC passes (looked up)
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"sweep_{r['forward_kernel']}_B{B}_{r['history']}")
            if rec:
                fa, wa = frames * (S * esz + S * 2), frames * (S * esz + S * 2 + 6)
                r["traffic"] = {"forward_hbm_bytes_per_launch": rec["forward_hbm_bytes_per_launch"], "backtrace_hbm_bytes_per_launch": rec["backtrace_hbm_bytes_per_launch"],
                                "forward_over_algorithmic": rec["forward_hbm_bytes_per_launch"] / fa,
                                "whole_path_over_algorithmic": (rec["forward_hbm_bytes_per_launch"] + rec["backtrace_hbm_bytes_per_launch"]) / wa,
                                "source": rec["profile"] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernels and shapes; looked up, not measured in this run)"}
        except Exception:
            pass
    sub = [0, min(B, 32) // 2, min(B, 32) - 1]
    rs, rl = __import__("oracle.viterbi_oracle", fromlist=["x"]).decode_c(
        logA_T, log_pi, E[sub].float().cpu().numpy(), lengths=None if lengths is None else lengths[sub].cpu().numpy())
    r["bit_exact_vs_oracle_sample"] = bool(np.array_equal(st[sub].cpu().numpy(), rs)) and bool(np.array_equal(ll[sub].cpu().numpy(), rl))
    if overlapped:
        try:
            wall, st2, ll2 = time_overlapped(dec, E, algo, steps=steps, lengths=lengths)
            r.update({"overlapped_ms_per_step": wall, "Mframes_per_s_overlapped": frames / wall / 1e3,
                      "overlapped_equals_serial": bool(torch.equal(st2, st) and torch.equal(ll2, ll))})
            del st2, ll2
        except torch.OutOfMemoryError:
            r["overlapped_ms_per_step"] = None       # two workspace slots do not fit beside the emissions
        # the schedule a caller should use for this batch: never one that is slower than its serial twin
        ov = r.get("overlapped_ms_per_step")
        r["best_schedule"] = "two streams" if ov is not None and ov < ser_ms else "one stream"
        r["Mframes_per_s_best"] = frames / min(ser_ms, ov if ov is not None else ser_ms) / 1e3
    del st, ll
    dec._ws = None
    dec._ws_slots = {}
    dec.set_option("reset", 0)
    torch.cuda.empty_cache()
    return r


def cu_masked_streams(dev):
    """Two HIP streams on disjoint halves of the chip's compute units (hipExtStreamCreateWithCUMask: CUs 0 .. n/2-1 and n/2 .. n-1),
    or None where the call is not available.  While one song per workgroup uses at most half the CUs, the forward stream on one half and
    everything else (emission builder, back-trace, voicing map) on the other keeps the builder's thousands of waves off the CUs that
    run the latency-bound forward workgroups (pipeline at B = 128: 11.76 -> 11.12 ms per step, scripts/pipeline_cumask.py)."""
    import ctypes
    try:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cus + 31) // 32
        if words < 2 or n_cus % 64:
            return None
        out = []
        for lo_half in (True, False):
            mask = [(0xFFFFFFFF if (w < words // 2) == lo_half else 0) for w in range(words)]
            st = ctypes.c_void_p()
            if hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), (ctypes.c_uint32 * words)(*mask)) != 0:
                return None
            out.append(torch.cuda.ExternalStream(st.value, device=dev))
        return out
    except (OSError, AttributeError):
        return None


def packed_row(dec, logA_T, log_pi, T, total_frames, dev, steps, seed=7):
    """A ragged batch as the reference sees it (recordings decoded whole, each with its own T): lengths uniform in [T/4, T], as
    many songs as hold `total_frames` frames, ONE packed [sum T_b, S] emission buffer, vit_decode_packed (forward slots packed
    longest-first on the host, back-trace chunks of equal length).  Whole decode timed with HIP events on the current stream."""
    S = dec.S
    rng = np.random.default_rng(seed)
    lens = []
    left = total_frames
    while left > 0:
        n = int(rng.integers(T // 4, T + 1))
        n = min(n, left)
        lens.append(n)
        left -= n
    lens = np.asarray(lens, np.int64)
    B = len(lens)
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    base = synth.emissions_peaks(32, T, S, seed=1234, device=dev)
    E = torch.empty((int(off[-1]), S), dtype=torch.float32, device=dev)
    for b in range(B):
        E[off[b]:off[b + 1]] = base[b % 32, :lens[b]]
    del base
    need = dec.workspace_bytes_packed(B, int(off[-1]))
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    st, ll = dec.decode_packed(E, off, out_dtype=torch.int32, workspace=ws)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        st, ll = dec.decode_packed(E, off, out_dtype=torch.int32, workspace=ws)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    sub = [0, B // 2, B - 1, int(np.argmin(lens))]
    from oracle import viterbi_oracle as vo
    ok = True
    for b in sub:
        rs, rl = vo.decode_c(logA_T, log_pi, E[off[b]:off[b + 1]].unsqueeze(0).cpu().numpy())
        ok = ok and bool(np.array_equal(st[off[b]:off[b + 1]].cpu().numpy(), rs[0])) and bool(ll[b].item() == rl[0])
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    r = {"songs": B, "frames_decoded": int(off[-1]), "lengths": f"uniform in [{T // 4}, {T}], mean {float(lens.mean()):.0f}, shortest {int(lens.min())}",
         "layout": "packed [sum T_b, S] emissions + host offsets (vit_decode_packed)", "forward_slots": int(min(B, 8 * n_cus)),
         "ms_per_step": ms, "Mframes_per_s": int(off[-1]) / ms / 1e3,
         "whole_path_hbm_frac": int(off[-1]) * (S * 4 + S * 2 + 6) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
         "workspace_GB": need / 1e9, "emissions_GB": E.numel() * 4 / 1e9, "bit_exact_vs_oracle_sample": ok}
    del E, ws, st, ll
    torch.cuda.empty_cache()
    return r


def checkpointed_row(dec, E, K, steps=3):
    """vit_decode_checkpointed on batch E: wall clock of whole calls, workspace, and equality with the normal decode."""
    B, T, _ = E.shape
    need = dec.workspace_bytes_checkpointed(B, T, K)
    ws = torch.empty(need + 256, dtype=torch.uint8, device=E.device)
    want_s, want_l = dec.decode(E, algo="banded", out_dtype=torch.int32)
    dec._ws = None
    torch.cuda.empty_cache()
    st, ll = dec.decode_checkpointed(E, segment_frames=K, out_dtype=torch.int32, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st, ll = dec.decode_checkpointed(E, segment_frames=K, out_dtype=torch.int32, workspace=ws)
    torch.cuda.synchronize()
    ck_ms = (time.perf_counter() - t0) / steps * 1e3
    r = {"songs": B, "segment_frames": K, "ms_per_step": ck_ms, "Mframes_per_s": B * T / ck_ms / 1e3,
         "workspace_GB": need / 1e9, "workspace_GB_full_history": dec.workspace_bytes(B, T) / 1e9,
         "equals_normal_decode": bool(torch.equal(st, want_s) and torch.equal(ll, want_l)),
         "note": f"wall clock of {steps} whole calls (pass 1 + {(T + K - 1) // K} segments x (forward, back-trace)), wave form"}
    del ws, st, ll, want_s, want_l
    torch.cuda.empty_cache()
    return r


def pipeline_block(dev, T, steps):
    """The post-processor callers actually run (Viterbi.__call__, tonet/for_paper.py:1817-1831): pitch logits [B, T, 360] ->
    observation log-probabilities (vit_obs_shaun: peak picking + soft voicing, tonet/for_paper.py:1733-1778) -> Viterbi decode ->
    voiced / bins (vit_voicing_map, :1828-1829), logits resident in HBM.  One stream: builder, forward, back-trace, map back to
    back (HIP events around each); overlapped: the builder + back-trace + map of step i on a second stream beside the forward pass
    of step i + 1 (two emission buffers, two workspace slots)."""
    from viterbi_spl_amd import emissions
    logA_T, log_pi = make_params("tonet", 361, 14)
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    out = {}
    # voicing of the synthetic logits: "toggle" = a voicing switch every other frame (the hardest case for the back-trace: a switch to
    # the unvoiced state has every voiced source as a candidate, so its row bound fails and the whole row is evaluated); "segments" =
    # voiced / unvoiced runs of ~120 frames, what a recording looks like
    for B, voicing in ((128, "toggle"), (1024, "toggle"), (1024, "segments")):
        key = f"B{B}" if voicing == "toggle" else f"B{B}_voiced_runs"
        X = synth.pitch_logits(min(B, 32), T, 360, seed=5, device=dev, voicing=voicing)
        if B > 32:
            X = X.repeat(B // 32, 1, 1).contiguous()
        E = [torch.empty((B, T, 361), dtype=torch.float32, device=dev) for _ in range(2)]
        st = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(2)]
        ll = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(2)]
        voiced = torch.empty((B, T), dtype=torch.uint8, device=dev)
        bins = torch.empty((B, T), dtype=torch.int32, device=dev)
        lib = __import__("viterbi_spl_amd._lib", fromlist=["x"]).load()

        def vmap(k):
            rc = lib.vit_voicing_map(st[k].data_ptr(), st[k].numel(), 360, voiced.data_ptr(), bins.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            assert rc == 0

        def one(k, ev=None):
            if ev:
                ev[0].record()
            emissions.shaun_log_emissions(X, out=E[k])
            if ev:
                ev[1].record()
            dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="forward", slot=k)
            if ev:
                ev[2].record()
            dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="backtrace", slot=k)
            if ev:
                ev[3].record()
            vmap(k)
            if ev:
                ev[4].record()

        one(0)
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
        for i in range(steps):
            one(0, evs[i])
        torch.cuda.synchronize()
        ph = [float(np.mean([e[j].elapsed_time(e[j + 1]) for e in evs])) for j in range(4)]
        total = float(np.mean([e[0].elapsed_time(e[4]) for e in evs]))
        ref_states = st[0].clone()
        bt_events = {k: round(v * 1000.0 / (B * T), 2) for k, v in dec.backtrace_counters(B, T).items() if k in ("span_misses", "whole_row_evaluations")}
        # parity of the decode inside the pipeline: the oracle on the emission rows the builder produced (three songs)
        from oracle import viterbi_oracle as vo
        sub = [0, min(B, 32) // 2, min(B, 32) - 1]
        rs, rl = vo.decode_c(logA_T, log_pi, E[0][sub].cpu().numpy())
        exact = bool(np.array_equal(st[0][sub].cpu().numpy(), rs)) and bool(np.array_equal(ll[0][sub].cpu().numpy(), rl))
        # overlapped: stream A = forward passes (critical path), stream B = builder of the next step, back-trace + map of the previous one
        sA, sB = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev)
        masked = cu_masked_streams(dev) if 2 * B <= torch.cuda.get_device_properties(dev).multi_processor_count else None
        if masked:                  # the forward workgroups on one half of the chip, builder / back-trace / map on the other
            sA, sB = masked
        dec.set_option("bt_chunks", dec.chunks_beside_forward(B))
        built = [None, None]
        fwd_done = [None, None]
        bt_done = [None, None]

        def run(n):
            with torch.cuda.stream(sB):
                emissions.shaun_log_emissions(X, out=E[0])
                built[0] = torch.cuda.Event()
                built[0].record()
            for i in range(n):
                k = i & 1
                with torch.cuda.stream(sA):
                    sA.wait_event(built[k])
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="forward", slot=k)
                    fwd_done[k] = torch.cuda.Event()
                    fwd_done[k].record()
                with torch.cuda.stream(sB):
                    if i + 1 < n:                             # emissions of step i + 1 (its buffer is free once back-trace i - 1 has run: same stream)
                        emissions.shaun_log_emissions(X, out=E[k ^ 1])
                        built[k ^ 1] = torch.cuda.Event()
                        built[k ^ 1].record()
                    sB.wait_event(fwd_done[k])
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="backtrace", slot=k)
                    vmap(k)
                    bt_done[k] = torch.cuda.Event()
                    bt_done[k].record()
            torch.cuda.synchronize()

        run(2)
        t0 = time.perf_counter()
        run(steps)
        ov = (time.perf_counter() - t0) / steps * 1e3
        same = bool(torch.equal(st[(steps - 1) & 1], ref_states))
        dec.set_option("bt_chunks", 0)
        frames = B * T
        bld_bytes = frames * (360 * 4 + 361 * 4)
        out[key] = {"songs": B, "voicing": voicing, "backtrace_events_per_1000_frames": bt_events, "builder_ms": ph[0], "forward_ms": ph[1], "backtrace_ms": ph[2], "voicing_map_ms": ph[3], "one_stream_ms_per_step": total,
                        "Mframes_per_s_one_stream": frames / total / 1e3, "overlapped_ms_per_step": ov, "Mframes_per_s_overlapped": frames / ov / 1e3,
                        "Mframes_per_s_best": frames / min(total, ov) / 1e3, "overlapped_equals_one_stream": same,
                        "overlapped_streams": "disjoint CU masks (hipExtStreamCreateWithCUMask): forward on one half of the chip, builder / back-trace / map on the other" if masked else "two plain streams",
                        "decode_bit_exact_vs_oracle_on_built_emissions": exact,
                        "builder": {"kernel": "observation_kernel (vit_obs_shaun, spw 5)", "Mframes_per_s": frames / ph[0] / 1e3,
                                    "roofline": {"bound": "hbm", "bytes_per_frame": 360 * 4 + 361 * 4, "achieved": bld_bytes / (ph[0] * 1e-3) / 1e9,
                                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bld_bytes / (ph[0] * 1e-3) / 1e9 / HBM_PEAK_GBS}},
                        "bytes_per_frame_logits_to_path": 360 * 4 + 361 * 4 + 361 * 4 + 361 * 2 + 6 + 5,
                        "voiced_fraction": float((voiced != 0).float().mean().item())}
        del X, E, st, ll, voiced, bins, ref_states
        dec._ws = None
        dec._ws_slots = {}
        torch.cuda.empty_cache()
    out["workload"] = (f"pitch logits [B, {T}, 360] fp32 resident in HBM (noise floor + melody-like bumps in the voiced frames, songs repeat with period 32; voicing \"toggle\": two frames of three voiced, \"segments\": runs of ~120 frames) -> vit_obs_shaun -> "
                       f"decode (tonet transition) -> vit_voicing_map; {steps} timed steps")
    return out


def extra_blocks(dev, args, headline_fwd_ms=None):
    """The saturation sweep and the high-resolution configuration, measured in this process after the headline."""
    out = {}
    T = args.frames
    NS = 10                     # timed steps per row
    # ---- sweep: S=361 fp32, tonet matrix, larger batches (the kernels the library picks for those sizes, and the alternatives)
    logA_T, log_pi = make_params("tonet", 361, 14)
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    sweep = {}
    for B in (1, 256, 512, 1024, 2048):       # B = 1 is BASELINE configs[1]: a single song, one workgroup, delta resident in LDS
        E = tiled_emissions(synth.emissions_peaks, B, T, 361, 1234, dev, torch.float32)
        sweep[f"B{B}"] = sweep_row(dec, logA_T, log_pi, E, "banded", NS)
        if B >= 1024:           # the wave form with the delta rows of even frames only (vit_plan_set_option "wave_history" = 2)
            sweep[f"B{B}_half_history"] = sweep_row(dec, logA_T, log_pi, E, "banded", NS, options={"wave_history": 2})
        del E
        torch.cuda.empty_cache()
    # data-dependent costs: dense i.i.d. emissions (SURVEY 8d's second distribution) and a ragged batch
    for B in (128, 1024):
        E = tiled_emissions(synth.emissions_dense, B, T, 361, 1234, dev, torch.float32)
        sweep[f"B{B}_dense_emissions"] = sweep_row(dec, logA_T, log_pi, E, "banded", NS, overlapped=False)
        del E
        torch.cuda.empty_cache()
    for B in (128, 1024):
        E = tiled_emissions(synth.emissions_peaks, B, T, 361, 1234, dev, torch.float32)
        g = torch.Generator().manual_seed(7)
        lengths = torch.randint(T // 4, T + 1, (B,), generator=g, dtype=torch.int64).to(dev)
        sweep[f"B{B}_ragged"] = sweep_row(dec, logA_T, log_pi, E, "banded", NS, lengths=lengths, overlapped=False)
        sweep[f"B{B}_ragged"]["lengths"] = f"uniform in [{T // 4}, {T}], mean {float(lengths.float().mean()):.0f}; block order (not sorted)"
        sweep[f"B{B}_ragged"]["forward_ms_if_all_full_length"] = sweep[f"B{B}"]["forward_ms"] if f"B{B}" in sweep else headline_fwd_ms if B == args.batch else None
        del E
        torch.cuda.empty_cache()
    # the reference's shipped parameters (msnet/viterbi_transition_matrix.dat, S = 321: the three-group wave variant)
    try:
        gp = np.load(os.path.join(ROOT, "tests", "golden", "params.npz"))
        mA, mpi = gp["msnet321_logA_T"], gp["msnet321_log_pi"]
        mdec = ViterbiDecoder(mA, mpi, dev)
        E = tiled_emissions(synth.emissions_peaks, 2048, T, 321, 1234, dev, torch.float32)
        sweep["B2048_msnet321"] = sweep_row(mdec, mA, mpi, E, "banded", NS, overlapped=False)
        sweep["B2048_msnet321"]["parameters"] = "msnet/viterbi_transition_matrix.dat + viterbi_init_probs.dat (payload in tests/golden/params.npz), S = 321"
        del E, mdec
        torch.cuda.empty_cache()
    except Exception as ex:            # (the fixture is part of the repo; a failure here must not cost the whole line)
        sweep["B2048_msnet321"] = {"error": repr(ex)}
    # a ragged batch as the reference sees it: packed layout + slot packing, as many frames as the uniform B = 2048 row
    try:
        sweep["B3277_ragged_packed"] = packed_row(dec, logA_T, log_pi, T, 2048 * T, dev, NS)
        if "B2048" in sweep:
            sweep["B3277_ragged_packed"]["vs_uniform_B2048_one_stream"] = sweep["B3277_ragged_packed"]["Mframes_per_s"] / sweep["B2048"]["Mframes_per_s"]
    except Exception as ex:
        sweep["B3277_ragged_packed"] = {"error": repr(ex)}
    # bounded-workspace decode (vit_decode_checkpointed): checkpoint rows in pass 1, segments of K frames re-run and back-traced
    for B, K in ((256, 1024), (1024, 1024), (2048, 1024)):
        E = tiled_emissions(synth.emissions_peaks, B, T, 361, 1234, dev, torch.float32)
        sweep[f"B{B}_checkpointed"] = checkpointed_row(dec, E, K)
        if f"B{B}" in sweep:
            sweep[f"B{B}_checkpointed"]["vs_normal_decode"] = sweep[f"B{B}_checkpointed"]["Mframes_per_s"] / sweep[f"B{B}"]["Mframes_per_s"]
        del E
        dec._ws = None
        torch.cuda.empty_cache()
    out["sweep"] = {"workload": f"T={T}, S=361, fp32 log-emissions (peaks unless the row says dense), tonet transition; songs repeat with period 32; "
                                f"{NS} timed steps per row: forward + back-trace back to back on one stream (forward_ms, backtrace_ms, Mframes_per_s) and "
                                "the two-stream schedule of the headline (overlapped_*); *_hbm_frac on algorithmic bytes (SURVEY 8d); "
                                "ragged rows count the frames actually decoded", **sweep}
    del dec
    # ---- configs[4]: S=722 (721 bins + unvoiced), fp16 emissions, 256 songs (and 512: two songs per CU cover each other's barriers)
    c4 = {}
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    for name, tr, dmax in (("jdc_band_dmax40", "tonet", 40), ("durrieu_dense", "durrieu", None)):
        A, pi = make_params(tr, 722, dmax or 14)
        dec = ViterbiDecoder(A, pi, dev)
        for B4 in (256, 512):
            E = tiled_emissions(synth.emissions_peaks, B4, T, 722, 1234, dev, torch.float16)
            r = sweep_row(dec, A, pi, E, "auto", NS, overlapped=True)
            if dec.info["banded_ok"]:
                r["valu_ceiling"] = valu_ceiling(dec, 722, B4 * T, r["forward_ms"])
            # issue ceiling of THIS kernel from its measured instruction mix (rocprofv3 --pmc SQ counters of the same kernel and shape,
            # profiles/r04_pmc_configs4.txt): vector instructions per frame and song spread over a CU's four SIMDs at one wave64
            # instruction per four cycles, and the LDS pipe's active cycles per frame; a CU works on one song's frame at a time
            prof = C4_ISSUE_PROFILE[name]
            cyc = max(prof["valu_instructions_per_frame_and_song"], prof["lds_active_cycles_per_frame_and_song"])
            peak = n_cus * 2.4e9 / cyc
            ic = {**prof, "cycles_per_frame_and_cu_at_the_ceiling": cyc, "peak_Mframes_per_s": peak / 1e6,
                  "achieved_Mframes_per_s": B4 * T / (r["forward_ms"] * 1e-3) / 1e6, "frac": B4 * T / (r["forward_ms"] * 1e-3) / peak}
            r["issue_ceiling"] = ic
            if not dec.info["banded_ok"]:   # the step-structured kernel shares its band maxima between targets: no per-candidate floor, the issue ceiling stands in
                r["valu_ceiling"] = ic
            c4[name if B4 == 256 else f"{name}_B{B4}"] = r
            del E
            torch.cuda.empty_cache()
        del dec
        torch.cuda.empty_cache()
    try:
        out["pipeline"] = pipeline_block(dev, T, NS)
    except Exception as ex:
        out["pipeline"] = {"error": repr(ex)}
    out["configs4"] = {"workload": f"[256, {T}, 722] fp16 log-emissions (peaks), songs repeat with period 32 (BASELINE configs[4]; the _B512 rows: 512 songs); "
                                   f"{NS} timed steps: one stream (forward_ms, backtrace_ms, Mframes_per_s) and the headline's two-stream schedule (overlapped_*)", **c4}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process has not touched the GPU; its N workers are children (no re-exec)
        raise SystemExit(launch_workers(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or plainly")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"     # the latter: rehearse the gather path on one GPU
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    B, T, S = args.batch, args.frames, args.states
    logA_T, log_pi = make_params(args.transition, S, args.dmax)
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    for kv in args.option:
        k, v = kv.split("=")
        dec.set_option(k, int(v))
    algo = args.algo
    if algo == "auto" and dec.info["banded_ok"]:
        algo = "banded"
    gen = synth.emissions_peaks if args.emissions == "peaks" else synth.emissions_dense
    dt = torch.float16 if args.f16 else torch.float32
    E = gen(B, T, S, seed=1234, device=dev, dtype=dt, first_song=rank * B)
    n_total = B * world
    NSLOT = 1 if (args.serial and not use_dist) else 2     # batches in flight
    for k in range(NSLOT):
        dec._workspace(B, T, k, algo)                        # allocate before anything is timed
    states_k = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(NSLOT)]
    loglik_k = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(NSLOT)]
    pipe = sharded.GatherPipeline(B, T, dev, n_slots=NSLOT) if use_dist else None    # gather buffers + handles per slot (rank 0 receives)
    bt_done = [None] * NSLOT        # event: back-trace of the batch in this slot finished
    s_fwd = torch.cuda.Stream(device=dev, priority=-1)        # the forward pass is the critical path: its workgroups go first
    s_bt = torch.cuda.Stream(device=dev) if not args.serial else s_fwd
    user_chunks = any(kv.split("=")[0] == "bt_chunks" for kv in args.option)
    if s_bt is not s_fwd and not user_chunks:
        dec.set_option("bt_chunks", dec.chunks_beside_forward(B))   # the back-trace on the units the forward pass leaves idle (0 = the library's count)

    def step(i, ev=None):
        k = pipe.acquire(i) if pipe is not None else i % NSLOT     # (waits for gather(i - NSLOT), which still reads states_k[k])
        with torch.cuda.stream(s_fwd):
            if bt_done[k] is not None and s_bt is not s_fwd:
                s_fwd.wait_event(bt_done[k])           # back-trace(i - NSLOT) still reads workspace slot k
            if ev is not None:
                ev[0].record()
            dec.decode_into(E, states_k[k], loglik_k[k], algo=algo, phase="forward", slot=k)
            if ev is not None:
                ev[1].record()
            fwd_done = torch.cuda.Event()
            fwd_done.record()
        with torch.cuda.stream(s_bt):
            if s_bt is not s_fwd:
                s_bt.wait_event(fwd_done)
            if ev is not None:
                ev[2].record()
            dec.decode_into(E, states_k[k], loglik_k[k], algo=algo, phase="backtrace", slot=k)
            if ev is not None:
                ev[3].record()
            bt_done[k] = torch.cuda.Event()
            bt_done[k].record()
            if pipe is not None:                       # ordered behind the back-trace (current stream), runs on the communicator's stream
                pipe.submit(k, states_k[k], loglik_k[k])
                if args.serial:
                    pipe.wait(k)

    def drain():
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize()

    def timed(n, with_events):
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)] if with_events else [None] * n
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i, evs[i])
        drain()
        if use_dist:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, evs

    ranks_info = None
    if use_dist:                       # who is in the job, as the communicator sees it (outside the timed region)
        mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "name": torch.cuda.get_device_name(dev),
                "songs": [rank * B, rank * B + B]}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
    for k in range(NSLOT):             # prime every slot once (code objects, first touch of the workspaces) whatever --warmup says
        step(k)
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    elapsed, events = timed(args.steps, True)
    states, loglik = states_k[(args.steps - 1) % NSLOT], loglik_k[(args.steps - 1) % NSLOT]
    fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
    bt_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))

    if rank == 0:
        esize = 2 if args.f16 else 4
        frames_per_launch = B * T
        bpf = S * esize + S * 2                                          # emission row in, uint16 back-pointer row out
        fwd_bytes = frames_per_launch * bpf
        value = n_total * T * args.steps / elapsed / 1e6
        achieved = fwd_bytes / (fwd_ms * 1e-3) / 1e9
        fwd_kernel = forward_kernel_name(dec, algo, B, S)
        traffic = bt_traffic = None
        traffic_source = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{fwd_kernel}_B{B}_T{T}_S{S}_{'f16' if args.f16 else 'f32'}"
        if os.path.exists(tf):       # PMC measurement of the same kernels on the same shapes (scripts/pmc_target.sh), not of this run
            try:
                rec = json.load(open(tf)).get(key, {})
                traffic = rec.get("forward_hbm_bytes_per_launch")
                bt_traffic = rec.get("backtrace_hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = f"profiles/traffic.json[{key}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernels and shapes ({rec.get('profile', 'r02')}); looked up, not measured in this run"
            except Exception:
                traffic = None
        workload = (f"batch of {B} songs per GPU, T={T}, S={S}, {'fp16' if args.f16 else 'fp32'} log-emissions ({args.emissions}), "
                    f"{args.transition} transition" + (f" (band half-width {args.dmax})" if args.transition == "tonet" else "") +
                    ", forward + back-trace" + (" + RCCL gather of paths" if use_dist else ""))
        if (B, T, S, args.f16, args.transition, args.dmax) == (128, 30000, 361, False, "tonet", 14):
            workload += " = BASELINE configs[2] (configs[3] at 8 GPUs)"
        out = {
            "metric": "Viterbi Mframes/s at S=361 T=30k; achieved HBM GB/s vs peak",
            "value": value, "unit": "Mframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "emission_storage": "f16" if args.f16 else "f32", "data": "synthetic",
            "config": {"workload": workload, "songs_per_gpu": B, "frames": T, "states": S, "emissions": args.emissions,
                       "emission_storage": "f16" if args.f16 else "f32",
                       "transition": args.transition, "band_half_width": args.dmax if args.transition == "tonet" else None,
                       "algo": algo, "forward_kernel": fwd_kernel, "plan": dec.info, "options": args.option},
            "schedule": ("one stream: forward, back-trace" + (", gather" if use_dist else "") + " of a step back to back") if args.serial else
                        ("two streams: back-trace of step i overlaps the forward pass of step i+1; two workspace slots" +
                         (f"; bt_chunks = {dec.chunks_beside_forward(B)} (the back-trace on the compute units the forward pass leaves idle)" if not user_chunks and dec.chunks_beside_forward(B) else "") +
                         ("; non-blocking gather on the communicator's stream" if use_dist else "")),
            "distributed": None if not use_dist else {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks_info,
                                                      "gathers_launched_on_rank0": pipe.launched, "gather": "non-blocking dist.gather of states [B,T] int32 + loglik [B] to rank 0 per step"},
            "roofline": {"bound": bound_of(fwd_kernel), "limited_by": limited_by(fwd_kernel, B), "kernel": fwd_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": fwd_bytes, "avg_launch_ms": fwd_ms, "bytes_per_frame": bpf,
                         "note": "algorithmic bytes per SURVEY 8d (emission row in + uint16 back-pointer row out); the kernels "
                                 "store the float32 delta row instead (lazy back-pointers, DESIGN.md 4.0)"},
            "kernels_ms": {"forward": fwd_ms, "backtrace": bt_ms},
            # the other ceiling of this recursion: one packed add + one max3 per two candidates and state is the minimum with
            # exact fp32 sums, and a SIMD issues one wave64 vector instruction per 4 cycles
            "valu_ceiling": valu_ceiling(dec, S, frames_per_launch, fwd_ms),
            "backtrace": {"algorithmic_bytes_per_launch": frames_per_launch * 6, "traffic": bt_traffic},
            "whole_path_bytes_per_frame": bpf + 6,
            "whole_path_hbm_frac": value * 1e6 * (bpf + 6) / 1e9 / (HBM_PEAK_GBS * world),
        }
        if world == 1 and not use_dist:
            torch.cuda.synchronize()
            if not args.no_traffic and (S, args.transition, args.dmax, args.f16, args.emissions) == (361, "tonet", 14, False, "peaks"):
                pm = measure_traffic(B, T, algo, args.option)
                if pm and fwd_kernel in pm:
                    bt_names = [k for k in pm if "backtrace" in k]
                    out["roofline"]["traffic"] = pm[fwd_kernel]["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate, no tracing "
                                                        "domains) over scripts/prof_target.py decoding the same batch shape twice; read bytes = 2 x FETCH_SIZE x 1024 "
                                                        "(gfx950: 64 B counted per 128-byte request), write bytes = WRITE_SIZE x 1024")
                    out["roofline"]["traffic_over_algorithmic"] = pm[fwd_kernel]["hbm_bytes_per_launch"] / fwd_bytes
                    out["backtrace"]["traffic"] = sum(pm[k]["hbm_bytes_per_launch"] for k in bt_names) if bt_names else None
                    out["pmc"] = pm
            if not args.serial:        # the same steps without the overlap, for the record (the library's own chunk count)
                if not user_chunks:
                    dec.set_option("bt_chunks", 0)
                ser, _, _ = time_serial(dec, E, algo, steps=3)
                out["serial_schedule"] = {**ser, "Mframes_per_s": B * T / ser["wall_ms_per_step"] / 1e3}
            if not args.no_cpu_baseline:
                cb, cbc = cpu_baseline(logA_T, log_pi, E, states, loglik, args.cpu_seconds)
                out["cpu_baseline"] = cb
                out["cpu_baseline_c"] = cbc
                out["gpu_over_cpu_1core"] = value / cb["value"]
            if not args.no_extras:
                del E, states_k, loglik_k, states, loglik
                dec._ws = None
                dec._ws_slots = {}
                torch.cuda.empty_cache()
                out.update(extra_blocks(dev, args, fwd_ms))
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
