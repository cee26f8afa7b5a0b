"""Experiment: forward pass and back-trace of consecutive steps on CU-masked streams (hipExtStreamCreateWithCUMask), so that
the back-trace of step i runs on the CUs the 128 one-song workgroups of step i+1 do not use.  Prints ms per step for
several mask layouts (timing only)."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.init()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
B, T = 128, 30000
E = synth.emissions_peaks(B, T, 361, seed=1234, device=dev)
st = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(2)]
ll = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(2)]
ALL = [0xFFFFFFFF] * 8
layouts = {
    "no masks": (None, None),
    "alternating bits (0101 / 1010)": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "low / high half of every word": ([0x0000FFFF] * 8, [0xFFFF0000] * 8),
    "first four / last four words": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "forward everywhere, back-trace on alternating bits": (ALL, [0xAAAAAAAA] * 8),
}
for name, (mf, mb) in layouts.items():
    sf = masked_stream(mf) if mf else torch.cuda.Stream(device=dev)
    sb = masked_stream(mb) if mb else torch.cuda.Stream(device=dev)
    done = [None, None]

    def run(n):
        for i in range(n):
            k = i % 2
            with torch.cuda.stream(sf):
                if done[k] is not None:
                    sf.wait_event(done[k])
                dec.decode_into(E, st[k], ll[k], algo="group", phase="forward", slot=k)
                fd = torch.cuda.Event()
                fd.record()
            with torch.cuda.stream(sb):
                sb.wait_event(fd)
                dec.decode_into(E, st[k], ll[k], algo="group", phase="backtrace", slot=k)
                done[k] = torch.cuda.Event()
                done[k].record()
        torch.cuda.synchronize()

    run(2)
    t0 = time.perf_counter()
    run(10)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name}: {ms:.3f} ms per step -> {B*T/ms/1e3:.1f} Mframes/s", flush=True)
