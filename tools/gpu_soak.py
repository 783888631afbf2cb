"""Soak check: the same 2^20-proof batch verified repeatedly (device-resident entry point, 1 and 2 streams via the env of the
process); every run must reproduce the generator's expected statuses bit for bit."""
import sys, importlib, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
n = 1 << 20
vk, proofs, inputs, exp = pkg.synth_groth16(0xB254AAAA, 2, n, invalid_every=16, agree=True, threads=16)
pvk = pkg.PreparedVk(vk); pvk.reserve(n)
dev = torch.device("cuda:0")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
st = torch.cuda.current_stream(dev)
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = time.time()
for r in range(reps):
    ds = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, st.cuda_stream)
    torch.cuda.synchronize()
    got = bytes(ds.cpu().numpy().tobytes())
    if got != exp:
        diff = [i for i in range(n) if got[i] != exp[i]]
        print("run %d: %d mismatches, first %s" % (r, len(diff), diff[:8]), flush=True)
        bad += 1
print("%d runs, %d bad, %.1f s" % (reps, bad, time.time() - t), flush=True)
sys.exit(1 if bad else 0)
