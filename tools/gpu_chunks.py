"""Does keeping the workspace inside the 256 MB Infinity Cache pay?  Same 2^20 proofs verified as 1, 4, 8 sub-batches."""
import sys, importlib, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
n = 1 << 20
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
pvk = pkg.PreparedVk(vk); pvk.reserve(n)
dev = torch.device("cuda:0")
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
ds = torch.zeros(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
for chunks in (1, 4, 8, 16, 1):
    m = n // chunks
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for c in range(chunks):
            pvk.verify_batch_device(dp.data_ptr() + 256 * m * c, di.data_ptr() + 64 * m * c, ds.data_ptr() + m * c, m, 256, 2, 0, st.cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    ok = bytes(ds.cpu().numpy().tobytes()) == exp
    print("chunks=%2d (m=%7d): %.1f ms  %.2f M proofs/s ok=%s" % (chunks, m, dt * 1e3, n / dt / 1e6, ok), flush=True)
