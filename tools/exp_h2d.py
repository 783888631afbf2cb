#!/usr/bin/env python3
"""Experiment: H2D bandwidth from pinned memory with the GPU idle and with the verification kernels running (is the copy a DMA or a starved blit kernel?)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
dev = torch.device("cuda:0")
n = 1 << 19
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540002, 2, n, invalid_every=16, agree=True, threads=16)
pvk = pkg.PreparedVk(vk); pvk.reserve(n, 0)
dp = torch.frombuffer(bytearray(proofs), dtype=torch.uint8).to(dev); di = torch.frombuffer(bytearray(inputs), dtype=torch.uint8).to(dev)
ds = torch.zeros(n, dtype=torch.uint8, device=dev)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)
src = torch.empty(320 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(320 << 20, dtype=torch.uint8, device=dev)
def copy_ms(pieces):
    sz = src.numel() // pieces
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    with torch.cuda.stream(side):
        for k in range(pieces):
            dst[k * sz:(k + 1) * sz].copy_(src[k * sz:(k + 1) * sz], non_blocking=True)
    side.synchronize()
    return (time.perf_counter() - t) * 1e3
for pieces in (1, 16):
    print("idle GPU, %2d pieces: %.2f ms = %.1f GB/s" % (pieces, copy_ms(pieces), 0.32 * 1.048576 / copy_ms(pieces) * 1e3))
for pieces in (1, 16):
    for _ in range(3):
        pvk.verify_batch_device(dp.data_ptr(), di.data_ptr(), ds.data_ptr(), n, 256, 2, 0, main.cuda_stream)
    ms = copy_ms(pieces)
    torch.cuda.synchronize(dev)
    print("busy GPU, %2d pieces: %.2f ms = %.1f GB/s" % (pieces, ms, 0.32 * 1.048576 / ms * 1e3))
