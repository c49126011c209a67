"""ctypes binding of libstarkperp.so (include/starkperp.h).  No fallbacks: if the library is
missing, or no MI355X is visible, every compute entry point raises."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# STARKPERP_LIB: another build of the SAME library (the sanitizer builds of csrc/Makefile, tools/run_sanitizers.sh);
# never a fallback - the path must exist or load() raises.
LIB_PATH = os.environ.get("STARKPERP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libstarkperp.so")

U64P = ctypes.POINTER(ctypes.c_uint64)
U8P = ctypes.POINTER(ctypes.c_uint8)


class StarkPerpError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_SIGNATURES = {
    "sp_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "sp_shutdown": (None, []),
    "sp_last_error": (ctypes.c_char_p, []),
    "sp_is_initialised": (ctypes.c_int, []),
    "sp_window_bits": (ctypes.c_int, []),
    "sp_table_bytes": (ctypes.c_size_t, []),
    "sp_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "sp_build_info": (ctypes.c_char_p, []),
    "sp_profile_begin": (ctypes.c_int, [ctypes.c_size_t]),
    "sp_profile_end": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_pedersen_batch": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t]),
    "sp_pedersen_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]),
    "sp_pedersen_point_batch": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t]),
    "sp_pedersen_chain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_pedersen_chain_right": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_pedersen_chains": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "sp_pedersen_chains_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_merkle_root": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "sp_merkle_build_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_merkle_forest_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p,
                                            ctypes.c_void_p]),
    "sp_merkle_sparse_root": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "sp_ntt_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]),
    "sp_lde_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                  ctypes.c_void_p, ctypes.c_void_p]),
    "sp_pedersen_trace_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "sp_air_eval_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_air_eval_shard_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                             ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]),
    "sp_fri_fold_shard_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                             ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "sp_air_eval_blocks_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                              ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_fri_fold_blocks_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                              ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_interpolate_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint,
                                          ctypes.c_void_p]),
    "sp_coset_eval_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "sp_rc16_trace_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_rc16_product_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p]),
    "sp_air_eval_rc16_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_felt_add_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p]),
    "sp_ec_ladder_trace_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "sp_air_eval_ec_ladder_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_init_devices": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    "sp_device_count": (ctypes.c_int, []),
    "sp_context_info": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_ecdsa_set_verify_policy": (ctypes.c_int, [ctypes.c_int]),
    "sp_ecdsa_get_verify_policy": (ctypes.c_int, []),
    "sp_range_check_trace_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_air_eval_range_check_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_ecdsa_trace_dev": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_air_eval_ecdsa_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_fri_fold_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "sp_commit_rows_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "sp_tree_create": (ctypes.c_int, [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_tree_create_on": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_order_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_tree_update": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sp_tree_get": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "sp_tree_root": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "sp_tree_destroy": (ctypes.c_int, [ctypes.c_int]),
    "sp_ecdsa_verify_batch": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t]),
    "sp_ecdsa_verify_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "sp_ecdsa_register_keys": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "sp_ecdsa_verify_keyed_dev": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    "sp_ecdsa_verify_batch_keyed": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t]),
    "sp_ecdsa_key_cache_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "sp_ecdsa_key_cache_reset": (ctypes.c_int, []),
    "sp_ecdsa_sign_batch": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t]),
    "sp_ecdsa_sign_rfc6979_batch": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t]),
    "sp_public_key_batch": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t]),
    "sp_ecdsa_sign_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "sp_ecdsa_sign_rfc6979_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "sp_public_key_batch_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]),
}


def declared_symbols():
    return sorted(_SIGNATURES)


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so with the same
    SONAME as /opt/rocm's; whichever is loaded first serves both.  If libstarkperp pulled in the
    system runtime first, a later `import torch` would bind its kernels to it and report
    "No HIP GPUs are available".  So when torch is installed, its runtime is loaded first
    (without importing torch); without torch the system runtime is used."""
    import importlib.util
    if os.environ.get("STARKPERP_SKIP_TORCH_RUNTIME"):  # processes that never touch torch (the CLI)
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the library and attach prototypes (does not touch the GPU)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise StarkPerpError(
                    "libstarkperp.so not built (%s); run `python __graft_entry__.py` or "
                    "`make -C stark-perpetual_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
            _preload_torch_hip_runtime()
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name, None)
                if fn is None:
                    continue  # symbol tests report what is missing
                fn.restype = res
                fn.argtypes = args
            _lib = lib
        return _lib


def last_error():
    return load().sp_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise StarkPerpError("%s failed (rc=%d): %s" % (what, rc, last_error()))


def ensure_init(device=None, window_bits=None):
    lib = load()
    if not lib.sp_is_initialised():
        explicit_device = device
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", os.environ.get("STARKPERP_DEVICE", "0")))
        if window_bits is None:
            window_bits = int(os.environ.get("STARKPERP_WINDOW_BITS", "0"))
        devices = os.environ.get("STARKPERP_DEVICES")  # "0,1,2,3": several contexts in this process (sp_init_devices)
        if devices and explicit_device is None:
            init_devices([int(v) for v in devices.split(",")], window_bits)
        else:
            check(lib.sp_init(device, window_bits), "sp_init")
    return lib


def init_devices(device_ids, window_bits=0):
    """Several devices in one process (include/starkperp.h sp_init_devices): context 0 = primary."""
    lib = load()
    ids = (ctypes.c_int * len(device_ids))(*device_ids)
    check(lib.sp_init_devices(len(device_ids), ids, int(window_bits or 0)), "sp_init_devices")
    return lib


def context_info(index):
    """(device, host-lane calls served) of context `index`."""
    dev, calls = ctypes.c_int(), ctypes.c_uint64()
    check(load().sp_context_info(index, ctypes.byref(dev), ctypes.byref(calls)), "sp_context_info")
    return dev.value, calls.value


# ---- felt marshalling -------------------------------------------------------------------------
MASK64 = (1 << 64) - 1


_INT_TO_BYTES = int.to_bytes


def pack_felts(values):
    """ints (0 <= v < 2^256) -> ctypes uint64 array of 4 LE limbs each."""
    n = len(values)
    if not n:
        return (ctypes.c_uint64 * 0)()
    try:  # plain ints: the unbound method skips one call per element
        raw = b"".join([_INT_TO_BYTES(v, 32, "little") for v in values])
    except TypeError:  # numpy integers and the like
        raw = b"".join([int(v).to_bytes(32, "little") for v in values])
    return (ctypes.c_uint64 * (4 * n)).from_buffer_copy(raw)


def unpack_felts(buf, n):
    raw = ctypes.string_at(buf, 32 * n)
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(n)]


def new_felts(n):
    return (ctypes.c_uint64 * (4 * n))()


def new_bytes(n):
    return (ctypes.c_uint8 * max(n, 1))()
