"""PlonK batches of 4096 with several calls in flight (host threads; one prepared key per thread and, second experiment, one shared key)."""
import importlib, json, os, random, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
base = [(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])) for f in fx.values() if f["variant"] == "plonk"]
rng = random.Random(4)
batch = 4096
proofs, inputs = [], []
for i in range(batch):
    p, q = base[i % len(base)]
    if i % 8 == 7:
        q = bytearray(q); q[rng.randrange(64)] ^= 1 << rng.randrange(8); q = bytes(q)
    proofs.append(p); inputs.append(q)
pb, ib = b"".join(proofs), b"".join(inputs)
ref = None
for shared in (False, True):
    for nt in (1, 2, 3, 4, 6):
        keys = [pkg.PreparedPlonkVk(vk) for _ in range(1 if shared else nt)]
        for k in keys: st = k.verify_batch(pb, ib)
        if ref is None: ref = st
        steps = 8
        out = [None] * nt
        def work(j):
            k = keys[0 if shared else j]
            for _ in range(steps): out[j] = k.verify_batch(pb, ib)
        th = [threading.Thread(target=work, args=(j,)) for j in range(nt)]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t
        assert all(o == ref for o in out)
        print(f"shared_key={shared} threads={nt}: {nt * steps * batch / dt / 1e3:8.1f} k proofs/s  ({dt * 1e3 / steps:6.2f} ms per round of {nt} batches)", flush=True)
        for k in keys: k.close()
