#!/usr/bin/env python3
"""Extract the reference's own test data into small committed fixtures.

Run once in the build container (needs /root/reference); the outputs are data, not source:
  * fixtures.json      -- for each of the 8 files in examples/binaries/: variant, the two decimal public
                          inputs, the raw gnark proof bytes (hex) and the vkey hash.  These are exactly the
                          values the reference's only test feeds to {Groth16,Plonk}Verifier::verify
                          (examples/script/src/main.rs:182-245).
  * plonk_vk.bin       -- the 34 368-byte gnark PlonK verifying key embedded in the committed guest ELF
                          examples/program/elf/plonk at file offset 315128; its
                          SHA-256 equals the plonk vkey hash stored in every PlonK fixture.
bincode layout (SP1 v2.0.0 SP1ProofWithPublicValues): u32 variant, then
String public_inputs[0], String public_inputs[1], String encoded_proof, String raw_proof (String = u64 LE
length + bytes), [u8;32] vkey_hash, ...
"""
import hashlib, json, os, struct, sys

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
PLONK_VK_OFFSET, PLONK_VK_LEN = 315128, 34368
PLONK_VK_SHA256 = "4aca240a3e5296e6a565f98dc728c6f48f8de4792a8fa365038c3b86952176f5"


def read_string(buf, off):
    (n,) = struct.unpack_from("<Q", buf, off)
    off += 8
    return buf[off:off + n].decode("ascii"), off + n


def parse_fixture(path):
    buf = open(path, "rb").read()
    (variant,) = struct.unpack_from("<I", buf, 0)
    off = 4
    pi0, off = read_string(buf, off)
    pi1, off = read_string(buf, off)
    encoded, off = read_string(buf, off)
    raw, off = read_string(buf, off)
    vkey_hash = buf[off:off + 32].hex()
    return {"variant": {2: "plonk", 3: "groth16"}[variant], "public_inputs": [pi0, pi1],
            "encoded_proof": encoded, "raw_proof": raw, "vkey_hash": vkey_hash}


def main():
    out = {}
    bdir = os.path.join(REF, "examples", "binaries")
    for fn in sorted(os.listdir(bdir)):
        if fn.endswith("_proof.bin"):
            out[fn[:-len("_proof.bin")]] = parse_fixture(os.path.join(bdir, fn))
    # the 8 fixture files themselves (21 KB of data): input of the SP1 fixture reader test
    import shutil
    os.makedirs(os.path.join(HERE, "sp1"), exist_ok=True)
    for fn in sorted(os.listdir(bdir)):
        if fn.endswith("_proof.bin"):
            shutil.copy(os.path.join(bdir, fn), os.path.join(HERE, "sp1", fn))
            os.chmod(os.path.join(HERE, "sp1", fn), 0o644)
    elf = open(os.path.join(REF, "examples", "program", "elf", "plonk"), "rb").read()
    vk = elf[PLONK_VK_OFFSET:PLONK_VK_OFFSET + PLONK_VK_LEN]
    assert hashlib.sha256(vk).hexdigest() == PLONK_VK_SHA256, "plonk vk hash mismatch"
    for name, fx in out.items():
        if fx["variant"] == "plonk":
            assert fx["vkey_hash"] == PLONK_VK_SHA256, name
    open(os.path.join(HERE, "plonk_vk.bin"), "wb").write(vk)
    json.dump(out, open(os.path.join(HERE, "fixtures.json"), "w"), indent=1, sort_keys=True)
    for name, fx in out.items():
        print(name, fx["variant"], len(fx["raw_proof"]) // 2, fx["vkey_hash"][:16])


if __name__ == "__main__":
    sys.exit(main())
