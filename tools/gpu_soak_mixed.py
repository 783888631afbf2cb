#!/usr/bin/env python3
"""Concurrency soak on one GPU: host threads that share prepared keys and hammer, side by side, the Groth16 batch entry at several sizes (cooperative
and one-proof-per-lane kernels), the single-proof entry (prepared-key cache, five keys through four slots), the RLC flag, a 300-input key (comb tables) and the PlonK batch entry
(thread pool, per-context scratch).  Every answer is compared with the generator's expected statuses / the first answer.
  python tools/gpu_soak_mixed.py [seconds]"""
import importlib, json, os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BN254_RLC_MIN_BATCH", "64")
import torch  # noqa: F401  (HIP runtime of torch first)
pkg = importlib.import_module("snark-bn254-verifier_amd")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
vk, proofs, inputs, exp = pkg.synth_groth16(0xB2540A01, 2, 40000, invalid_every=8, agree=True, threads=16)
pvk = pkg.PreparedVk(vk)
keys = [pkg.synth_groth16(0xB2540B00 + k, 2, 4, invalid_every=2, agree=True, threads=2) for k in range(5)]
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
pl_vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
base = [(bytes.fromhex(f["raw_proof"]), b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])) for f in fx.values() if f["variant"] == "plonk"]
rng = random.Random(5)
pp, pi = [], []
for i in range(5200):
    p, q = base[i % len(base)]
    if i % 5 == 4:
        q = bytearray(q); q[rng.randrange(64)] ^= 1 << rng.randrange(8); q = bytes(q)
    pp.append(p); pi.append(q)
ppb, pib = b"".join(pp), b"".join(pi)
wide = pkg.synth_groth16(0xB2540C00, 300, 3000, invalid_every=9, agree=True, threads=16)     # comb tables, chunked MSM, eight-lane reduction
wide_pvk = pkg.PreparedVk(wide[0])
plonk = pkg.PreparedPlonkVk(pl_vk)
plonk_ref = plonk.verify_batch(ppb, pib)
assert plonk_ref.count(bytes([pkg.ACCEPT])) == 4160
stop = time.time() + secs
errors, counts, lock = [], {}, threading.Lock()


def note(kind):
    with lock:
        counts[kind] = counts.get(kind, 0) + 1


def g16_batches(seed):
    r = random.Random(seed)
    while time.time() < stop and not errors:
        n = r.choice([1, 7, 300, 4096, 9000, 25000, 40000])
        off = r.randrange(0, 40000 - n + 1)
        flags = r.choice([0, 0, pkg.FLAG_RLC])
        st = pvk.verify_batch(proofs[256 * off:256 * (off + n)], inputs[64 * off:64 * (off + n)], n, flags=flags)
        if st != exp[off:off + n]:
            errors.append(("g16 batch", n, off, flags))
        note("g16_batch")


def g16_single(seed):
    r = random.Random(seed)
    while time.time() < stop and not errors:
        vk_k, pr, inp, ex = keys[r.randrange(5)]
        i = r.randrange(4)
        ins = [int.from_bytes(inp[64 * i + 32 * j:64 * i + 32 * j + 32], "big") for j in range(2)]
        if pkg.Groth16Verifier.verify(pr[256 * i:256 * i + 256], vk_k, ins) != ex[i]:
            errors.append(("g16 single", i))
        note("g16_single")


def g16_wide(seed):
    r = random.Random(seed)
    vk_w, pr, inp, ex = wide
    while time.time() < stop and not errors:
        n = r.choice([1, 40, 777, 3000])
        off = r.randrange(0, 3000 - n + 1)
        st = wide_pvk.verify_batch(pr[256 * off:256 * (off + n)], inp[9600 * off:9600 * (off + n)], n)
        if st != ex[off:off + n]:
            errors.append(("g16 wide", n, off))
        note("g16_wide")


def plonk_batches(seed):
    r = random.Random(seed)
    while time.time() < stop and not errors:
        if r.random() < 0.05:
            # a large call with the pairing checks batched across proofs (BN254_FLAG_RLC: one pass of 10 400 proofs = the 5200 twice), beside the small ones
            st = plonk.verify_batch(ppb * 2, pib * 2, 10400, flags=pkg.FLAG_RLC if r.random() < 0.7 else 0)
            if st != plonk_ref * 2:
                errors.append(("plonk rlc batch", 10400, 0))
            note("plonk_rlc_batch")
            continue
        n = r.choice([1, 33, 700, 1500, 2500, 4500, 5000, 5200])      # 2500 / 4500 / 5000: the window-table limits of a context
        off = r.randrange(0, 5200 - n + 1)
        st = plonk.verify_batch(ppb[904 * off:904 * (off + n)], pib[64 * off:64 * (off + n)], n)
        if st != plonk_ref[off:off + n]:
            errors.append(("plonk batch", n, off))
        note("plonk_batch")


th = [threading.Thread(target=g16_batches, args=(1,)), threading.Thread(target=g16_batches, args=(2,)), threading.Thread(target=g16_single, args=(3,)), threading.Thread(target=g16_wide, args=(6,)),
      threading.Thread(target=plonk_batches, args=(4,)), threading.Thread(target=plonk_batches, args=(5,)), threading.Thread(target=plonk_batches, args=(7,))]   # three callers on ONE PlonK key: its context pool
def progress():
    # a line a minute: a run that stays silent for minutes looks hung to whoever watches it
    while time.time() < stop and not errors:
        time.sleep(min(60.0, max(0.1, stop - time.time())))
        with lock:
            print(json.dumps({"elapsed_s": round(secs - (stop - time.time())), "calls": dict(counts)}), file=sys.stderr, flush=True)


th.append(threading.Thread(target=progress))
for t in th: t.start()
for t in th: t.join()
print(json.dumps({"seconds": secs, "calls": counts, "errors": errors[:5], "all_ok": not errors}))
sys.exit(1 if errors else 0)
