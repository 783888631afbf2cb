import sys, importlib, ctypes as C, random
sys.path.insert(0, '.')
pkg = importlib.import_module("snark-bn254-verifier_amd")
from oracle import oracle as O
L = pkg.lib()
rng = random.Random(4)
g1 = O.g1_mul(O.g1_gen(), rng.randrange(1, O.R)); g2 = O.g2_mul(O.g2_gen(), rng.randrange(1, O.R))
n = 66000
out = (C.c_uint8 * (384 * n))()
assert L.bn254_dbg_pairing(g1 * n, g2 * n, out, C.c_size_t(n), 0) == 0, L.bn254_last_error()
out = bytes(out); ref = out[:384]
bad = [i for i in range(n) if out[384 * i:384 * i + 384] != ref]
print("pairing ref ok:", ref == O.pairing(g1, g2), "bad lanes:", len(bad), bad[:10], flush=True)
fl = (C.c_uint8 * n)()
assert L.bn254_dbg_g2_subgroup(g2 * n, fl, C.c_size_t(n), 0) == 0
fl = bytes(fl); bad = [i for i in range(n) if fl[i] != 1]
print("subgroup bad lanes:", len(bad), bad[:10], flush=True)
