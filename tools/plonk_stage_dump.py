#!/usr/bin/env python3
"""Where do the device and the host versions of PlonK stage 1 part?  Needs a diagnostics build of the library (EXTRA=-DBN254_PLONK_MARKS; BN254_LIB_PATH
selects it): runs the reference's first PlonK fixture as a one-proof batch on the device, runs stage 1 of the same proof on the host, and compares the
intermediate values both sides dumped (bn254_plonk.hpp::PL_DUMP)."""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["gamma", "beta", "alpha", "zeta", "zeta^n", "zeta^n - 1", "acc", "den0", "den1", "den2", "den3", "pre3", "acc_inv", "inv0", "inv1", "inv2", "inv3", "lagrange_one",
         "pi (inputs)", "pi (+bsb)", "a2l1", "cl (before pi)", "cl (final)", "claimed0", "claimed1", "claimed5", "zs_value",
         "digest gamma", "digest beta", "digest alpha", "digest zeta"]


def main():
    pkg = importlib.import_module("snark-bn254-verifier_amd")
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
    vk = open(os.path.join(ROOT, "tests", "golden", "plonk_vk.bin"), "rb").read()
    f = [f for f in fx.values() if f["variant"] == "plonk"][0]
    proof = bytes.fromhex(f["raw_proof"]); inputs = b"".join(int(x).to_bytes(32, "big") for x in f["public_inputs"])
    L = pkg.lib()
    pvk = pkg.PreparedPlonkVk(vk)
    st = pvk.verify_batch(proof, inputs, 1)
    dev = (C.c_uint8 * 2048)(); host = (C.c_uint8 * 2048)(); hs = C.c_int()
    assert L.bn254_dbg_plonk_dump_device(dev) == 0
    L.bn254_dbg_plonk_dump_host.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
    assert L.bn254_dbg_plonk_dump_host(pvk._h, proof, len(proof), inputs, 2, host, C.byref(hs)) == 0
    print("device status", st[0], "host stage-1 status", hs.value)
    first = None
    for k, name in enumerate(NAMES):
        d = int.from_bytes(bytes(dev[32 * k:32 * k + 32]), "little"); h = int.from_bytes(bytes(host[32 * k:32 * k + 32]), "little")
        ok = d == h
        if not ok and first is None:
            first = name
        print("%-16s %s  dev %064x%s" % (name, "==" if ok else "!=", d, "" if ok else "\n%22shost %064x" % ("", h)))
    print("first difference:", first)
    # the SHA-256 compressions of stage 1: message words as the compression function read them, state after
    dd = (C.c_uint32 * (32 * 24))(); hd = (C.c_uint32 * (32 * 24))(); dn = C.c_uint32(); hn = C.c_uint32()
    assert L.bn254_dbg_plonk_sha_dump_device(dd, C.byref(dn)) == 0 and L.bn254_dbg_plonk_sha_dump_host(hd, C.byref(hn)) == 0
    print("compressions: device %d host %d (the host's first transcript continues from the key's saved state; the device's too)" % (dn.value, hn.value))
    for j in range(min(dn.value, hn.value, 32)):
        dm, hm = list(dd[24 * j:24 * j + 16]), list(hd[24 * j:24 * j + 16]); ds, hs2 = list(dd[24 * j + 16:24 * j + 24]), list(hd[24 * j + 16:24 * j + 24])
        print("block %2d message %s state %s" % (j, "==" if dm == hm else "!=", "==" if ds == hs2 else "!="))
        if dm != hm:
            print("   dev  msg", " ".join("%08x" % x for x in dm)); print("   host msg", " ".join("%08x" % x for x in hm))
        if ds != hs2:
            print("   dev  st ", " ".join("%08x" % x for x in ds)); print("   host st ", " ".join("%08x" % x for x in hs2))
        if dm != hm or ds != hs2:
            break


if __name__ == "__main__":
    main()
