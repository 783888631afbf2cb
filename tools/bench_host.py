#!/usr/bin/env python3
"""The host-buffer entry (bn254_groth16_verify_batch on pageable memory) against the device-resident one, batch 2^20; BN254_HOST_TIMING=1 prints
where the host thread spends the call."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
pvk = pkg.PreparedVk(vk)
dev = torch.device("cuda:0")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
ds = torch.zeros(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
pvk.reserve(n, 0)
out = {}
for it in range(4):
    if it == 1:
        torch.cuda.synchronize(dev); t = time.perf_counter()
    pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, st.cuda_stream)
torch.cuda.synchronize(dev)
out["resident_ms"] = (time.perf_counter() - t) / 3 * 1e3
assert bytes(ds.cpu().numpy().tobytes()) == exp
pvk.verify_batch(proofs, inputs, n)
t = time.perf_counter()
for it in range(3):
    s = pvk.verify_batch(proofs, inputs, n)
out["host_ms"] = (time.perf_counter() - t) / 3 * 1e3
assert s == exp
out["ratio"] = out["resident_ms"] / out["host_ms"]
print(json.dumps(out))
