"""ctypes binding of libbn254_verify_amd.so.  Mirrors the reference's API names (verifier/src/lib.rs:29-49):
Groth16Verifier.verify(proof, vk, public_inputs) plus the new verify_batch."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REJECT, ACCEPT, ERR_NOT_MEMBER, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP, ERR_INPUT_LEN, ERR_MALFORMED = range(7)
VK_REFERENCE, VK_GNARK = 0, 1
FLAG_STRICT_SCALARS, FLAG_RLC = 1, 2
RAW_PROOF_LEN = 324
NUM_KERNELS = 4
ABI_VERSION = 5   # include/bn254_verify.h: BN254_ABI_VERSION


class Bn254Error(RuntimeError):
    pass


def lib_path():
    # BN254_LIB_PATH: a diagnostics build of the same library (tools/plonk_stage_marks.py); never set in tests or the bench
    return os.environ.get("BN254_LIB_PATH") or os.path.join(HERE, "libbn254_verify_amd.so")


def build(verbose=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j2"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return lib_path()


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process, whatever the import order.  The library needs `libamdhip64.so.7` (soname); a PyTorch-ROCm wheel bundles
    its own copy of that runtime and asks for it as `libamdhip64.so` (no version), so the dynamic loader only recognises the two requests as
    the same object when torch's copy is mapped FIRST.  Mapped the other way round the process ends up with two runtimes and the second one
    sees no devices (the library then answers BN254_E_NO_DEVICE).  So when a torch with a bundled runtime is installed and not yet imported,
    its copy is mapped here, before the library -- without importing torch.  A host without torch (the C++ / Rust case) is not affected."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path()):
            raise Bn254Error("libbn254_verify_amd.so is not built (run __graft_entry__.build()); there is no fallback path")
        _share_torch_hip_runtime()
        L = C.CDLL(lib_path())
        L.bn254_last_error.restype = C.c_char_p
        L.bn254_version.restype = C.c_char_p
        L.bn254_status_string.restype = C.c_char_p
        L.bn254_groth16_kernel_name.restype = C.c_char_p
        L.bn254_groth16_vk_num_public.restype = C.c_size_t
        L.bn254_synth_groth16_vk_len.restype = C.c_size_t
        L.bn254_groth16_vk_prepare.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p)]
        L.bn254_groth16_vk_free.argtypes = [C.c_void_p]
        L.bn254_groth16_vk_num_public.argtypes = [C.c_void_p]
        L.bn254_groth16_verify_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_uint]
        L.bn254_groth16_verify_batch_multi.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint64, C.c_uint]
        L.bn254_groth16_proof_write_raw.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p]
        L.bn254_groth16_verify_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
        L.bn254_groth16_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.bn254_plonk_last_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_size_t)]
        L.bn254_groth16_rlc_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint)]
        L.bn254_groth16_verify.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint, C.c_void_p]
        L.bn254_groth16_last_kernel_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.bn254_synth_groth16.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bn254_synth_groth16_vk_len.argtypes = [C.c_size_t]
        L.bn254_synth_groth16_range.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bn254_shard_plan.argtypes = [C.c_size_t, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.bn254_plonk_vk_prepare.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.bn254_plonk_vk_free.argtypes = [C.c_void_p]
        L.bn254_plonk_vk_num_public.argtypes = [C.c_void_p]
        L.bn254_plonk_vk_num_public.restype = C.c_size_t
        L.bn254_plonk_verify_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int]
        L.bn254_plonk_verify_batch_flags.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_uint]
        L.bn254_plonk_verify.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p]
        L.bn254_plonk_verify_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_uint]
        L.bn254_plonk_verify_batch_multi.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint64, C.c_uint]
        L.bn254_plonk_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.bn254_plonk_footprint.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        if L.bn254_abi_version() != ABI_VERSION:
            raise Bn254Error("libbn254_verify_amd.so has ABI revision %d, this binding was written for %d" % (L.bn254_abi_version(), ABI_VERSION))
        L.bn254_groth16_kernel_kind_name.restype = C.c_char_p
        L.bn254_groth16_kernel_kind_name.argtypes = [C.c_int]
        L.bn254_set_profile_kernels.argtypes = [C.c_uint]
        L.bn254_set_rlc_params.argtypes = [C.c_long, C.c_int, C.c_long]
        L.bn254_set_rlc_params.restype = None
        L.bn254_groth16_kernel_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_size_t)]
        L.bn254_groth16_kernel_profile_all.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def kernel_kinds():
    return [lib().bn254_groth16_kernel_kind_name(i).decode() for i in range(lib().bn254_groth16_num_kernel_kinds())]


def set_profile_kernels(names=None):
    """Select the kernel kinds whose launches get HIP events (None = all)."""
    kinds = kernel_kinds()
    mask = 0xffffffff if names is None else sum(1 << kinds.index(x) for x in names)
    lib().bn254_set_profile_kernels(mask)


def set_rlc_params(min_batch=-1, adaptive=-1, share_min_lanes=-1):
    """Knobs of FLAG_RLC (bn254_set_rlc_params; -1 leaves a knob alone): batch size from which the flag is honoured, adaptive bypass, lanes a
    launch part must keep for shared Miller-loop accumulators."""
    lib().bn254_set_rlc_params(min_batch, adaptive, share_min_lanes)


def set_plonk_params(piece=-1, workers=-1, big_from=-1, big_piece=-1):
    """Knobs of the PlonK batch plan (bn254_set_plonk_params; -1 leaves a knob alone)."""
    L = lib()
    L.bn254_set_plonk_params.argtypes = [C.c_long, C.c_int, C.c_long, C.c_long]
    L.bn254_set_plonk_params.restype = None
    L.bn254_set_plonk_params(piece, workers, big_from, big_piece)


def _check(rc):
    if rc != 0:
        raise Bn254Error("bn254 error %d: %s" % (rc, lib().bn254_last_error().decode()))


def _inputs_bytes(public_inputs):
    return b"".join(x if isinstance(x, (bytes, bytearray)) else int(x).to_bytes(32, "big") for x in public_inputs)


class PreparedPlonkVk:
    """Opaque prepared PlonK verifying key (bn254_plonk_vk_prepare)."""

    def __init__(self, vk_bytes):
        self._h = C.c_void_p()
        _check(lib().bn254_plonk_vk_prepare(bytes(vk_bytes), len(vk_bytes), C.byref(self._h)))
        self.n_public = lib().bn254_plonk_vk_num_public(self._h)

    def verify_batch(self, proofs, public_inputs, n=None, proof_stride=904, n_public=None, device=0, flags=0):
        """proofs: n * proof_stride bytes; public_inputs: n * n_public * 32 bytes.  Returns n status bytes.  flags: FLAG_RLC batches the pairing checks of a pass across
        proofs (honoured from 8192 proofs per pass; exact fallback on the groups that fail)."""
        n_public = self.n_public if n_public is None else n_public
        if n is None:
            n = len(proofs) // proof_stride
        st = (C.c_uint8 * max(n, 1))()
        _check(lib().bn254_plonk_verify_batch_flags(self._h, bytes(proofs), proof_stride, bytes(public_inputs), n_public, n, st, device, flags))
        return bytes(st)[:n]

    def verify_batch_multi(self, proofs, public_inputs, device_mask, n=None, proof_stride=904, n_public=None, flags=0):
        """Same over the GPUs selected by the bits of device_mask (contiguous shards, one host thread per device)."""
        n_public = self.n_public if n_public is None else n_public
        if n is None:
            n = len(proofs) // proof_stride
        st = (C.c_uint8 * max(n, 1))()
        _check(lib().bn254_plonk_verify_batch_multi(self._h, bytes(proofs), proof_stride, bytes(public_inputs), n_public, n, st, device_mask, flags))
        return bytes(st)[:n]

    def verify_batch_device(self, d_proofs, d_inputs, d_status, n, proof_stride=904, n_public=None, device=0, stream=None, flags=0):
        """Raw device pointers (ints).  Host-synchronous: waits for `stream`, returns when the status bytes are in d_status."""
        n_public = self.n_public if n_public is None else n_public
        _check(lib().bn254_plonk_verify_batch_device(self._h, d_proofs, proof_stride, d_inputs, n_public, n, d_status, device, stream, flags))

    def reserve(self, n, proof_stride=0, device=0):
        """Allocate now what a batch of up to n proofs needs (proof_stride > 0: also the pinned staging of the host-buffer entry)."""
        _check(lib().bn254_plonk_reserve(self._h, n, proof_stride, device))

    def footprint(self, device=0):
        """(bytes of device memory this key's contexts hold on `device`, contexts that hold any)"""
        b, c = C.c_size_t(0), C.c_int(0)
        _check(lib().bn254_plonk_footprint(self._h, device, C.byref(b), C.byref(c)))
        return b.value, c.value

    def last_timing(self, device=0):
        """Stage and kernel durations (ms) of the first sub-batch of the last verify_batch (bn254_plonk_last_timing)."""
        ms = (C.c_float * 9)()
        lanes = (C.c_size_t * 2)()
        _check(lib().bn254_plonk_last_timing(self._h, device, ms, lanes))
        names = ("host_copy", "k_plonk_stage1", "k_g1_msm_rows_digest", "k_g1_sum_affine_digest", "k_plonk_stage2", "k_g1_msm_rows_kzg", "k_g1_sum_affine_kzg", "pairing_check", "sub_batch_wall")
        return dict(zip(names, ms)), (lanes[0], lanes[1])

    def close(self):
        if self._h:
            lib().bn254_plonk_vk_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlonkVerifier:
    """Mirror of the reference's `PlonkVerifier::verify(proof, vk, public_inputs)` (verifier/src/lib.rs:69-73): returns the
    status byte (ACCEPT or an error code; PlonK never answers REJECT)."""

    @staticmethod
    def verify(proof, vk, public_inputs):
        st = C.c_uint8(0xEE)
        ib = _inputs_bytes(public_inputs)
        _check(lib().bn254_plonk_verify(bytes(proof), len(proof), bytes(vk), len(vk), ib, len(ib) // 32, C.byref(st)))
        return st.value


class PreparedVk:
    """Opaque prepared verifying key (bn254_groth16_vk_prepare)."""

    def __init__(self, vk_bytes, mode=VK_REFERENCE):
        self._h = C.c_void_p()
        _check(lib().bn254_groth16_vk_prepare(bytes(vk_bytes), len(vk_bytes), mode, C.byref(self._h)))
        self.n_public = lib().bn254_groth16_vk_num_public(self._h)

    @property
    def handle(self):
        return self._h

    def verify_batch(self, proofs, public_inputs, n=None, proof_stride=256, n_public=None, device=0, flags=0):
        """proofs: bytes (n * proof_stride); public_inputs: bytes (n * n_public * 32). Returns n status bytes.
        flags: FLAG_STRICT_SCALARS | FLAG_RLC (include/bn254_verify.h)."""
        n_public = self.n_public if n_public is None else n_public
        if n is None:
            n = len(proofs) // proof_stride
        st = (C.c_uint8 * max(n, 1))()
        _check(lib().bn254_groth16_verify_batch(self._h, bytes(proofs), proof_stride, bytes(public_inputs), n_public, n, st, device, flags))
        return bytes(st)[:n]

    def verify_batch_multi(self, proofs, public_inputs, device_mask, n=None, proof_stride=256, n_public=None, flags=0):
        """Same over the GPUs selected by the bits of device_mask (contiguous shards, one host thread per device)."""
        n_public = self.n_public if n_public is None else n_public
        if n is None:
            n = len(proofs) // proof_stride
        st = (C.c_uint8 * max(n, 1))()
        _check(lib().bn254_groth16_verify_batch_multi(self._h, bytes(proofs), proof_stride, bytes(public_inputs), n_public, n, st, device_mask, flags))
        return bytes(st)[:n]

    def verify_batch_device(self, d_proofs, d_inputs, d_status, n, proof_stride=256, n_public=None, device=0, stream=None, flags=0):
        """Raw device pointers (ints); enqueues on `stream` (a hipStream_t value) and returns (FLAG_RLC: after one stream sync)."""
        n_public = self.n_public if n_public is None else n_public
        _check(lib().bn254_groth16_verify_batch_device(self._h, d_proofs, proof_stride, d_inputs, n_public, n, d_status, device, stream, flags))

    def reserve(self, n, device=0):
        _check(lib().bn254_groth16_reserve(self._h, n, device))

    def rlc_state(self, device=0):
        """(share of the checked proofs the recent FLAG_RLC passes sent to the exact fallback, -1.0 before the first pass; calls that bypassed the mode)."""
        share, by = C.c_float(), C.c_uint()
        _check(lib().bn254_groth16_rlc_state(self._h, device, C.byref(share), C.byref(by)))
        return share.value, by.value

    def last_kernel_ms(self, device=0):
        ms = (C.c_float * NUM_KERNELS)()
        _check(lib().bn254_groth16_last_kernel_ms(self._h, device, ms))
        return {lib().bn254_groth16_kernel_name(i).decode(): ms[i] for i in range(NUM_KERNELS)}

    def kernel_profile(self, device=0):
        """Per kernel kind: {name: (launches, total_ms)} of the last profiled batch, and the proofs each launch covered."""
        k = lib().bn254_groth16_num_kernel_kinds()
        cnt = (C.c_uint * k)(); ms = (C.c_float * k)(); per = C.c_size_t(0)
        _check(lib().bn254_groth16_kernel_profile(self._h, device, cnt, ms, C.byref(per)))
        return {lib().bn254_groth16_kernel_kind_name(i).decode(): (int(cnt[i]), float(ms[i])) for i in range(k) if cnt[i]}, int(per.value)

    def kernel_profile_all(self, device=0):
        """Per kernel kind over the first two sub-batches (two streams): {name: (launches, total_ms, union_ms)}, and the proofs per launch."""
        k = lib().bn254_groth16_num_kernel_kinds()
        cnt = (C.c_uint * k)(); ms = (C.c_float * k)(); un = (C.c_float * k)(); per = C.c_size_t(0)
        _check(lib().bn254_groth16_kernel_profile_all(self._h, device, cnt, ms, un, C.byref(per)))
        return {lib().bn254_groth16_kernel_kind_name(i).decode(): (int(cnt[i]), float(ms[i]), float(un[i])) for i in range(k) if cnt[i]}, int(per.value)

    def close(self):
        if self._h:
            lib().bn254_groth16_vk_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Groth16Verifier:
    """Mirror of the reference's `Groth16Verifier` (verifier/src/lib.rs:29-49)."""

    @staticmethod
    def verify(proof, vk, public_inputs, mode=VK_REFERENCE):
        """Returns the status byte: ACCEPT = Ok(true), REJECT = Ok(false), ERR_INPUT_LEN = Err(PrepareInputsFailed);
        the other ERR_* codes are the reference's panics (unwrap of a loader error)."""
        st = C.c_uint8(0xEE)
        ib = _inputs_bytes(public_inputs)
        _check(lib().bn254_groth16_verify(bytes(proof), len(proof), bytes(vk), len(vk), ib, len(public_inputs), mode, C.byref(st)))
        return st.value

    @staticmethod
    def verify_batch(proofs, vk, public_inputs, mode=VK_REFERENCE, device=0):
        """proofs: list of byte strings; public_inputs: list of lists. Returns a list of status bytes."""
        pvk = PreparedVk(vk, mode)
        try:
            n = len(proofs)
            stride = max([256] + [len(p) for p in proofs])
            pb = b"".join(bytes(p).ljust(stride, b"\0") for p in proofs)
            npub = len(public_inputs[0]) if n else 0
            ib = b"".join(_inputs_bytes(x) for x in public_inputs)
            short = [len(p) < 256 for p in proofs]
            st = list(pvk.verify_batch(pb, ib, n, stride, npub, device))
            return [ERR_MALFORMED if s else v for s, v in zip(short, st)]
        finally:
            pvk.close()


def proof_write_raw(a, b, c):
    """The 324-byte raw gnark proof (A | B | C | no commitments | zero PoK) that groth16/converter.rs:14-26 reads."""
    out = (C.c_uint8 * RAW_PROOF_LEN)()
    _check(lib().bn254_groth16_proof_write_raw(bytes(a), bytes(b), bytes(c), out))
    return bytes(out)


def shard_plan(n, device_mask, device_count):
    """[(device, first, count)] of bn254_groth16_verify_batch_multi's shards (bn254_shard_plan: host arithmetic, no GPU)."""
    devs = (C.c_int * 64)(); first = (C.c_size_t * 64)(); cnt = (C.c_size_t * 64)(); k = C.c_int(0)
    _check(lib().bn254_shard_plan(n, device_mask, device_count, devs, first, cnt, C.byref(k)))
    return [(devs[i], first[i], cnt[i]) for i in range(k.value)]


def synth_groth16(seed, n_public, n, invalid_every=16, agree=True, threads=0, l_identity=False, first=0):
    """Deterministic synthetic gnark-format workload: (vk, proofs, inputs, expected_status) as bytes.  first: global index of the first
    proof (proof i of the stream depends on (seed, i) only, so a rank can generate its own shard)."""
    L = lib()
    vk = (C.c_uint8 * L.bn254_synth_groth16_vk_len(n_public))()
    proofs = (C.c_uint8 * max(256 * n, 1))()
    inputs = (C.c_uint8 * max(32 * n_public * n, 1))()
    exp = (C.c_uint8 * max(n, 1))()
    # l_identity: every proof with index = 3 mod 7 gets public inputs that make its public-input point L the identity (a valid proof)
    _check(L.bn254_synth_groth16_range(seed, n_public, first, n, invalid_every, (1 if agree else 0) | (2 if l_identity else 0), threads, vk, proofs, inputs, exp))
    return bytes(vk), bytes(proofs)[:256 * n], bytes(inputs)[:32 * n_public * n], bytes(exp)[:n]
