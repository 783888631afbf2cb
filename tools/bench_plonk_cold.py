import importlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("snark-bn254-verifier_amd")
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
fx = json.load(open("tests/golden/fixtures.json"))
vk = open("tests/golden/plonk_vk.bin", "rb").read()
f = [f for f in fx.values() if f["variant"] == "plonk"][0]
pp, pi = bytes.fromhex(f["raw_proof"]), [int(x) for x in f["public_inputs"]]
t = time.perf_counter(); st = pkg.PlonkVerifier.verify(pp, vk, pi); a = time.perf_counter() - t
assert st == pkg.ACCEPT
t = time.perf_counter()
for _ in range(20): st = pkg.PlonkVerifier.verify(pp, vk, pi)
b = (time.perf_counter() - t) / 20
print(json.dumps({"plonk_verify_first_call_with_a_key_ms": round(a * 1e3, 2), "plonk_verify_cached_key_ms": round(b * 1e3, 3)}))
