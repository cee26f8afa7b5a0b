"""Child process of tests/test_gpu_parity.py::test_sharded_decode_over_rccl_single_rank (test infrastructure).

One rank, backend "nccl" (= RCCL on ROCm), started as a FRESH process: ViterbiDecoder + sharded.decode_sharded (blocking
gather) and sharded.gather_paths_async (non-blocking gather on the communicator's stream, shards landing in place)
against the CPU oracle.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import viterbi_oracle as vo  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, sharded, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
    dec = ViterbiDecoder(A, pi, dev)
    B, T = 6, 500
    E = synth.emissions_peaks(B, T, 361, seed=99, device=dev)
    ref_s, ref_l = vo.decode_c(A, pi, E.cpu().numpy())
    out = {}
    # blocking path: decode this rank's block, one gather of states + loglik to rank 0
    st, ll = sharded.decode_sharded(lambda e: dec.decode(e, out_dtype=torch.int32), E, n_songs=B, dst=0)
    out["blocking"] = bool(np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l))
    # non-blocking path of bench.py: back-trace on the current stream, gather on the communicator's stream, in place
    ok = True
    for algo in ("group", "wave"):
        states = torch.empty((B, T), dtype=torch.int32, device=dev)
        loglik = torch.empty((B,), dtype=torch.float32, device=dev)
        out_s = torch.full((1, B, T), -7, dtype=torch.int32, device=dev)
        out_l = torch.zeros((1, B), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            dec.decode_into(E, states, loglik, algo=algo)
            works = sharded.gather_paths_async(states, loglik, out_s, out_l, dst=0)
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        ok = ok and bool(np.array_equal(out_s[0].cpu().numpy(), ref_s) and np.array_equal(out_l[0].cpu().numpy(), ref_l))
    out["async_in_place"] = ok
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
