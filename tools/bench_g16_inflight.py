"""Groth16 batches of 4096 with several calls in flight: k prepared handles of the same key, k streams, one host thread enqueues round-robin."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vk, proofs, inputs, expected = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
dev = torch.device("cuda", 0)
d_proofs = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev)
d_inputs = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
for k in (1, 2, 3, 4, 6):
    pvks = [pkg.PreparedVk(vk, pkg.VK_REFERENCE) for _ in range(k)]
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    sts = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(k)]
    for p in pvks: p.reserve(n, 0)
    def rnd():
        for j in range(k):
            pvks[j].verify_batch_device(d_proofs.data_ptr(), d_inputs.data_ptr(), sts[j].data_ptr(), n, 256, 2, 0, streams[j].cuda_stream)
    rnd(); torch.cuda.synchronize(dev)
    rounds = 20
    t = time.perf_counter()
    for _ in range(rounds): rnd()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t
    assert all(bytes(s.cpu().numpy().tobytes()) == expected for s in sts)
    print(f"n={n} handles={k}: {k * rounds * n / dt / 1e6:6.3f} M proofs/s ({dt * 1e3 / rounds:6.2f} ms per round of {k})", flush=True)
    for p in pvks: p.close()
