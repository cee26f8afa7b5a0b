"""Experiment: the logits -> path pipeline at B = 128 with the builder / back-trace stream and the forward stream on disjoint CU masks
(hipExtStreamCreateWithCUMask): does keeping the emission builder's 8192 waves off the 128 CUs that run the latency-bound forward
workgroups win back the 1.7 ms they cost it?  (timing only; run on the GPU box)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, emissions, synth, _lib  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.init()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


A, pi = bench.make_params("tonet", 361, 14)
dec = ViterbiDecoder(A, pi, dev)
lib = _lib.load()
B, T = 128, 30000
X = synth.pitch_logits(32, T, 360, seed=5, device=dev).repeat(B // 32, 1, 1).contiguous()
E = [torch.empty((B, T, 361), dtype=torch.float32, device=dev) for _ in range(2)]
st = [torch.empty((B, T), dtype=torch.int32, device=dev) for _ in range(2)]
ll = [torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(2)]
voiced = torch.empty((B, T), dtype=torch.uint8, device=dev)
bins = torch.empty((B, T), dtype=torch.int32, device=dev)
ALL = [0xFFFFFFFF] * 8
layouts = {
    "no masks": (None, None),
    "low / high half of every word": ([0x0000FFFF] * 8, [0xFFFF0000] * 8),
    "alternating bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "first four / last four words": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "forward everywhere, builder on the high halves": (ALL, [0xFFFF0000] * 8),
    "first four words / last two words (64 CUs)": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 6 + [0xFFFFFFFF] * 2),
    "first four words / last word (32 CUs)": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 7 + [0xFFFFFFFF]),
    # three streams: the builder throttled to a few CUs so that its HBM traffic spreads over the whole forward pass
    "three streams: forward 4 words / back-trace 3 words / builder 1 word": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 3 + [0], [0] * 7 + [0xFFFFFFFF]),
    "three streams: forward 4 words / back-trace 2 words / builder 2 words": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 2 + [0] * 2, [0] * 6 + [0xFFFFFFFF] * 2),
}
for chunks in (dec.chunks_beside_forward(B), 0):
    for name, masks in layouts.items():
        mf, mb = masks[0], masks[1]
        sA = masked_stream(mf) if mf else torch.cuda.Stream(device=dev, priority=-1)
        sB = masked_stream(mb) if mb else torch.cuda.Stream(device=dev)
        sC = masked_stream(masks[2]) if len(masks) > 2 else None
        dec.set_option("bt_chunks", chunks)
        built, fwd_done, bt_done = [None, None], [None, None], [None, None]

        def run3(n):
            with torch.cuda.stream(sC):
                emissions.shaun_log_emissions(X, out=E[0])
                built[0] = torch.cuda.Event(); built[0].record()
            for i in range(n):
                k = i & 1
                with torch.cuda.stream(sA):
                    sA.wait_event(built[k])
                    if bt_done[k] is not None: sA.wait_event(bt_done[k])          # workspace slot k is free
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="forward", slot=k)
                    fwd_done[k] = torch.cuda.Event(); fwd_done[k].record()
                if i + 1 < n:
                    with torch.cuda.stream(sC):
                        if fwd_done[k ^ 1] is not None: sC.wait_event(fwd_done[k ^ 1])   # forward i - 1 has read E[k ^ 1]
                        emissions.shaun_log_emissions(X, out=E[k ^ 1])
                        built[k ^ 1] = torch.cuda.Event(); built[k ^ 1].record()
                with torch.cuda.stream(sB):
                    sB.wait_event(fwd_done[k])
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="backtrace", slot=k)
                    assert lib.vit_voicing_map(st[k].data_ptr(), st[k].numel(), 360, voiced.data_ptr(), bins.data_ptr(), torch.cuda.current_stream(dev).cuda_stream) == 0
                    bt_done[k] = torch.cuda.Event(); bt_done[k].record()
            torch.cuda.synchronize()

        def run(n):
            with torch.cuda.stream(sB):
                emissions.shaun_log_emissions(X, out=E[0])
                built[0] = torch.cuda.Event(); built[0].record()
            for i in range(n):
                k = i & 1
                with torch.cuda.stream(sA):
                    sA.wait_event(built[k])
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="forward", slot=k)
                    fwd_done[k] = torch.cuda.Event(); fwd_done[k].record()
                with torch.cuda.stream(sB):
                    if i + 1 < n:
                        emissions.shaun_log_emissions(X, out=E[k ^ 1])
                        built[k ^ 1] = torch.cuda.Event(); built[k ^ 1].record()
                    sB.wait_event(fwd_done[k])
                    dec.decode_into(E[k], st[k], ll[k], algo="banded", phase="backtrace", slot=k)
                    assert lib.vit_voicing_map(st[k].data_ptr(), st[k].numel(), 360, voiced.data_ptr(), bins.data_ptr(), torch.cuda.current_stream(dev).cuda_stream) == 0
            torch.cuda.synchronize()

        if sC is not None:
            run = run3
        run(2)
        built, fwd_done, bt_done = [built[0], built[1]], [None, None], [None, None]
        t0 = time.perf_counter()
        run(10)
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"bt_chunks {chunks:2d}  {name}: {ms:.3f} ms per step -> {B * T / ms / 1e3:.1f} Mframes/s", flush=True)
