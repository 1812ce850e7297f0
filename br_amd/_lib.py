"""ctypes binding of libbrx.so (include/brx.h).  No torch types cross this boundary.

The library is the product path: if it is missing or no GPU is usable, every compute entry
raises -- there is no CPU fallback and nothing here ever touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRX_LIB_PATH") or os.path.join(_HERE, "lib", "libbrx.so")  # override: A/B of builds

BRX_OK = 0
BRX_ERR_ARG = -1
BRX_ERR_OVERFLOW = -7
BRX_ERR_NODEVICE = -4
BRX_ERR_UNSUPPORTED = -6

METHOD_IDS = {"one": 0, "two": 1, "graph": 2, "greedy": 3, "gap_size": 4, "gap-size": 4}
COUNT_AUTO, COUNT_DENSE, COUNT_SORTED = 0, 1, 2


class BrxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"brx status {status}: {msg}")
        self.status = status


class Method(C.Structure):
    _fields_ = [("method", C.c_uint8), ("confirm", C.c_uint8), ("max_search", C.c_uint8)]


class Synth(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("read_len", C.c_uint32),
                ("sub_e4", C.c_uint32), ("ins_e4", C.c_uint32), ("del_e4", C.c_uint32)]


_vp, _u8p, _u64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/brx.h one to one
SIGNATURES = {
    "brx_strerror": (C.c_char_p, [C.c_int]),
    "brx_last_error": (C.c_char_p, []),
    "brx_version": (C.c_int, []),
    "brx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "brx_profile_enable": (C.c_int, [C.c_int]),
    "brx_profile_reset": (C.c_int, []),
    "brx_profile_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), _u64p]),
    "brx_profile_names": (C.c_int, [C.c_char_p, C.c_size_t]),
    "brx_set_new": (C.c_int, [C.c_uint8, C.c_int, _pp]),
    "brx_set_new_from_solid_bytes": (C.c_int, [_vp, C.c_size_t, C.c_int, _pp]),
    "brx_set_insert_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "brx_set_set": (C.c_int, [_vp, C.c_uint64, C.c_bool]),
    "brx_set_get": (C.c_bool, [_vp, C.c_uint64]),
    "brx_set_get_batch": (C.c_int, [_vp, _vp, C.c_uint32, _vp]),
    "brx_set_k": (C.c_uint8, [_vp]),
    "brx_set_device": (C.c_int, [_vp]),
    "brx_set_export_solid_bytes": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "brx_set_popcount": (C.c_int, [_vp, _u64p]),
    "brx_set_sparse": (C.c_int, [_vp]),
    "brx_set_bits_state": (C.c_int, [_vp]),
    "brx_set_device_bits": (C.c_int, [_vp, _pp, _u64p]),
    "brx_set_extract_keys_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64, _u64p, _vp]),
    "brx_set_or_keys_device": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "brx_set_index_build": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "brx_set_index_build_from_keys_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, _vp]),
    "brx_set_keylist_device": (C.c_int, [_vp, _pp, _u64p, _vp]),
    "brx_set_fingerprint": (C.c_int, [_vp, _u64p, _vp]),
    "brx_set_index_drop": (C.c_int, [_vp]),
    "brx_set_index_info": (C.c_int, [_vp, _u64p]),
    "brx_set_get_batch_indexed": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _u64p]),
    "brx_set_free": (None, [_vp]),
    "brx_set_count_begin": (C.c_int, [C.c_uint8, C.c_int, C.c_int, _pp]),
    "brx_set_count_add_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "brx_set_count_add_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp]),
    "brx_set_count_finish": (C.c_int, [_vp, C.c_uint8, _vp, _pp]),
    "brx_set_count_finish_into": (C.c_int, [_vp, C.c_uint8, _vp, _vp]),
    "brx_counter_reset": (C.c_int, [_vp, _vp]),
    "brx_counter_spectrum": (C.c_int, [_vp, _u64p, _vp]),
    "brx_counter_load_counts": (C.c_int, [_vp, C.c_uint64, C.c_char_p, C.c_uint64]),
    "brx_counter_device_counts": (C.c_int, [_vp, _pp, _u64p]),
    "brx_counter_clamp": (C.c_int, [_vp, C.c_uint8, _vp]),
    "brx_counter_l1_view": (C.c_int, [_vp, _pp, _pp, C.POINTER(C.c_uint32), _u64p]),
    "brx_counter_add_partitioned_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "brx_counter_free": (None, [_vp]),
    "brx_comm_unique_id": (C.c_int, [_vp]),
    "brx_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _pp]),
    "brx_comm_init_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), _pp]),
    "brx_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "brx_exchange_build_partitioned": (C.c_int, [_vp, _vp, C.c_uint8, _vp, _vp]),
    "brx_exchange_reduce_counts": (C.c_int, [_vp, _vp, C.c_uint8, _vp]),
    "brx_exchange_plan": (C.c_int, [_u64p, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), _u64p, _u64p, _u64p, _u64p]),
    "brx_comm_last_stats": (C.c_int, [_vp, _u64p]),
    "brx_comm_free": (None, [_vp]),
    "brx_chain_new": (C.c_int, [_vp, C.POINTER(Method), C.c_uint32, C.c_bool, _pp]),
    "brx_chain_correct_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(_u8p), C.POINTER(_u64p)]),
    "brx_chain_correct_batch_async": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "brx_chain_correct_batch_wait": (C.c_int, [_vp, C.POINTER(_u8p), C.POINTER(_u64p)]),
    "brx_chain_correct_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp, C.c_uint64, _vp,
                                                 _u64p, _vp]),
    "brx_chain_last_stats": (C.c_int, [_vp, _u64p]),
    "brx_chain_free": (None, [_vp]),
    "brx_buf_free": (None, [_vp]),
    "brx_host_alloc": (_vp, [C.c_size_t]),
    "brx_host_free": (None, [_vp]),
    "brx_devpool_trim": (None, []),
    "brx_devpool_bytes": (C.c_uint64, []),
    "brx_run_correction_fd": (C.c_int, [_vp, C.POINTER(Method), C.c_uint32, C.c_bool, C.c_int, C.c_int, C.c_uint32, _u64p]),
    "brx_count_fasta_fd": (C.c_int, [_vp, C.c_int, C.c_uint32, _u64p]),
    "brx_set_insert_fasta_fd": (C.c_int, [_vp, C.c_int, C.c_uint32, _u64p]),
    "brx_set_insert_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _vp]),
    "brx_synth_genome_device": (C.c_int, [C.POINTER(Synth), C.c_int, _vp, _vp]),
    "brx_synth_reads_device": (C.c_int, [C.POINTER(Synth), C.c_int, _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64,
                                         _vp, _u64p, _vp]),
    "brx_synth_genome_host": (C.c_int, [C.POINTER(Synth), _vp]),
    "brx_synth_reads_host": (C.c_int, [C.POINTER(Synth), _vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, _vp, _u64p]),
}

_lib = None


def _adopt_torch_runtime() -> None:
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so.7 (+ HSA
    runtime); libbrx.so needs `libamdhip64.so.7` by soname.  If libbrx is loaded first it pulls
    /opt/rocm's copy, torch later loads its bundled copy next to it, and the second runtime in
    the process sees no GPU.  Importing torch first makes the loader resolve libbrx's dependency
    to the copy torch already mapped.  A host that never uses torch in-process (a Rust/C++
    caller of the C ABI, or BRX_NO_TORCH=1) simply gets /opt/rocm's runtime via RUNPATH."""
    import sys
    if "torch" in sys.modules or os.environ.get("BRX_NO_TORCH") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """Load libbrx.so.  Raises (loudly) if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    _adopt_torch_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C br_amd/csrc` (or __graft_entry__.build()). "
            "br_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(status: int) -> None:
    if status != BRX_OK:
        L = lib()
        msg = (L.brx_last_error() or b"").decode(errors="replace") or L.brx_strerror(status).decode()
        raise BrxError(status, msg)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().brx_device_count(C.byref(n)))
    return n.value


def profile_enable(on: bool = True) -> None:
    check(lib().brx_profile_enable(1 if on else 0))


def profile_reset() -> None:
    check(lib().brx_profile_reset())


def profile_get(name: str):
    ms, n = C.c_double(0), C.c_uint64(0)
    check(lib().brx_profile_get(name.encode(), C.byref(ms), C.byref(n)))
    return ms.value, n.value


def profile_all() -> dict:
    buf = C.create_string_buffer(4096)
    check(lib().brx_profile_names(buf, 4096))
    names = [s for s in buf.value.decode().split(",") if s]
    return {nm: dict(zip(("total_ms", "launches"), profile_get(nm))) for nm in names}
