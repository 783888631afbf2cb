#!/usr/bin/env python3
"""Batch sizes around the hand-over from the cooperative kernels to the one-proof-per-lane kernels (inputs resident, statuses checked)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
dev = torch.device("cuda:0")
nmax = 65536
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540001, 2, nmax, invalid_every=16, agree=True, threads=16)
pvk = pkg.PreparedVk(vk); pvk.reserve(nmax, 0)
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
st = torch.cuda.current_stream(dev)
for n in (16384, 20480, 20481, 24576, 28672, 30720, 32768, 40960, 40961, 49152, 65536):
    ds = torch.zeros(n, dtype=torch.uint8, device=dev)
    reps = 5
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize(dev); t = time.perf_counter()
        pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, st.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t) / reps
    assert bytes(ds.cpu().numpy().tobytes()) == exp[:n], n
    print(json.dumps({"batch": n, "ms_per_batch": round(dt * 1e3, 3), "proofs_per_s": round(n / dt)}), flush=True)
