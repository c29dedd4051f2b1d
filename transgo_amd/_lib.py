"""ctypes loader for libtransgo_hip.so (the C ABI declared in include/transgo_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no GPU is visible when a
context is created, the caller gets an exception."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtransgo_hip.so")     # the in-tree build only: no environment override of the product library


class TgConfig(ctypes.Structure):
    _fields_ = [("board_size", ctypes.c_int32), ("encode_dim", ctypes.c_int32), ("max_step", ctypes.c_int32),
                ("komi", ctypes.c_float), ("n_games", ctypes.c_int32), ("num_simulation", ctypes.c_int32),
                ("parallel_readouts", ctypes.c_int32), ("wu_loss", ctypes.c_int32), ("c_puct1", ctypes.c_double),
                ("c_puct2", ctypes.c_double), ("arena_slots", ctypes.c_int32), ("net_blocks", ctypes.c_int32),
                ("net_filters", ctypes.c_int32), ("device", ctypes.c_int32), ("net_precision", ctypes.c_int32),
                ("record_games", ctypes.c_int32), ("pool_slots", ctypes.c_int32), ("reserved", ctypes.c_int32 * 5)]


class TgMt19937(ctypes.Structure):
    _fields_ = [("key", ctypes.c_uint32 * 624), ("pos", ctypes.c_int32)]


class TransgoError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/transgo_hip.h declares
_vp, _i32p, _u8p, _f32p = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
SIGNATURES = {
    "tg_version": (ctypes.c_int, []),
    "tg_config_default": (None, [ctypes.POINTER(TgConfig)]),
    "tg_create": (ctypes.c_int, [ctypes.POINTER(TgConfig), ctypes.POINTER(ctypes.c_void_p)]),
    "tg_destroy": (None, [_vp]),
    "tg_last_error": (ctypes.c_char_p, [_vp]),
    "tg_sync": (ctypes.c_int, [_vp]),
    "tg_state_size": (ctypes.c_int, [_vp]),
    "tg_env_reset": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "tg_env_step": (ctypes.c_int, [_vp, _vp, _vp, _i32p, ctypes.c_int, _u8p, _u8p]),
    "tg_env_query": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _u8p, _u8p, _f32p, _f32p, _f32p, _i32p, _i32p, _u8p]),
    "tg_env_show": (ctypes.c_int, [_vp, _vp]),
    "tg_sp_reset": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_sp_reset_from": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_sp_root_states": (ctypes.c_int, [_vp, _vp]),
    "tg_sp_batch_rows": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "tg_sp_batch_obs": (ctypes.c_int, [_vp, _vp, ctypes.c_int32]),
    "tg_sp_set_eval": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32]),
    "tg_sp_eval": (ctypes.c_int, [_vp]),
    "tg_sp_expand_roots": (ctypes.c_int, [_vp]),
    "tg_sp_begin_move": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "tg_sp_collect": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "tg_sp_absorb": (ctypes.c_int, [_vp]),
    "tg_sp_search": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "tg_sp_root_info": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "tg_sp_draw_uniform": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_sp_rng_state": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(TgMt19937)]),
    "tg_sp_rng_get": (ctypes.c_int, [_vp, _vp]),
    "tg_sp_rng_set": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_sp_play": (ctypes.c_int, [_vp, _vp, _vp]),
    "tg_sp_final": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "tg_sp_game_errors": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _vp]),
    "tg_sp_tree_truncations": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "tg_sp_pool_stats": (ctypes.c_int, [_vp] + [ctypes.POINTER(ctypes.c_uint64)] * 4),
    "tg_sp_finished": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "tg_sp_harvest": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "tg_sp_stats": (ctypes.c_int, [_vp] + [ctypes.POINTER(ctypes.c_uint64)] * 4 + [ctypes.POINTER(ctypes.c_int32)] * 2),
    "tg_net_blob_floats": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "tg_net_load": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int]),
    "tg_net_blob_floats_arch": (ctypes.c_size_t, [ctypes.c_int] * 3 + [ctypes.c_char_p]),
    "tg_net_load_arch": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t, ctypes.c_int]),
    "tg_net_load_async": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t]),
    "tg_net_load_async_dev": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_size_t]),
    "tg_net_load_poll": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "tg_net_predict": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp]),
    "tg_net_range": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float)]),
    "tg_prof_enable": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "tg_prof_read": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]),
    "tg_prof_skipped": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "tg_prof_enable_tree": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "tg_prof_read_tree": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(ctypes.c_uint64)]),
    "tg_replay_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "tg_replay_destroy": (None, [_vp]),
    "tg_replay_append": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "tg_replay_append_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "tg_replay_info": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int)]),
    "tg_replay_sample": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "tg_host_mt_seed": (None, [ctypes.POINTER(TgMt19937), ctypes.c_uint32]),
    "tg_host_mt_next32": (ctypes.c_uint32, [ctypes.POINTER(TgMt19937)]),
    "tg_host_mt_random_sample": (ctypes.c_double, [ctypes.POINTER(TgMt19937)]),
    "tg_host_mt_choice_index": (ctypes.c_int32, [ctypes.POINTER(TgMt19937), ctypes.c_int32]),
    "tg_host_sub_encode": (None, [ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "tg_host_mt_dirichlet": (ctypes.c_int, [ctypes.POINTER(TgMt19937), ctypes.c_double, ctypes.c_int32, _vp]),
}


def load():
    """Load the library (once).  Raises TransgoError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libtransgo_hip links the system one under the
    # same SONAME, so whichever is mapped first serves both.  torch cannot enumerate devices through the system runtime
    # ("No HIP GPUs are available"), the library runs fine on torch's -- so when torch is installed it is imported first.
    # (A plain-C host, examples/c_host_min.c, never loads torch and uses the system runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise TransgoError(f"{LIB_PATH} not found: build it with `make -C transgo_amd/csrc` "
                           "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != 0:
        msg = lib.tg_last_error(ctx)
        raise TransgoError(f"libtransgo_hip error {rc}: {msg.decode() if msg else '?'}")


def default_config():
    cfg = TgConfig()
    load().tg_config_default(ctypes.byref(cfg))
    return cfg


class Context:
    """Owns one tg_ctx."""

    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        h = ctypes.c_void_p()
        rc = self.lib.tg_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            msg = self.lib.tg_last_error(None)
            raise TransgoError(f"tg_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h
        self.state_size = self.lib.tg_state_size(h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        check(self.lib, self.h, getattr(self.lib, name)(self.h, *args))
