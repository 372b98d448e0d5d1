"""ctypes binding of csrc/librdf_hip.so (the C ABI declared in include/rdf_hip.h).

This is the only place the shared library is opened.  There is no CPU fallback: a missing
library, or a machine without a HIP device, raises.
"""
import ctypes
import os

from . import _build

_c_int, _c_float, _c_void_p, _c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/rdf_hip.h declares
SIGNATURES = {
    "rdf_eval_forest": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int,
                                 _c_void_p, _c_int, _c_void_p, _c_int, _c_float, _c_void_p]),
    "rdf_eval_tree": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_void_p,
                               _c_void_p]),
    "rdf_composite": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p,
                               _c_void_p]),
    "rdf_layered_run": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                 _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                 _c_int, _c_float, _c_void_p]),
    "rdf_layered_run_hand": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                      _c_int, _c_float, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "rdf_forest_packed_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "rdf_forest_pack": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "rdf_eval_forest_packed_stats": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                                              _c_int, _c_void_p, _c_void_p]),
    "rdf_forest_set_deep_from": (_c_int, [_c_void_p, _c_int]),
    "rdf_forest_info": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int),
                                 ctypes.POINTER(_c_float)]),
    "rdf_forest_forget": (_c_int, [_c_void_p]),
    "rdf_forest_tune": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int,
                                 _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_eval_forest_packed": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                        _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p]),
    "rdf_eval_forest_packed_filled": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                               _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p]),
    "rdf_eval_forest_packed_split": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                              _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int,
                                              _c_int, ctypes.POINTER(_c_int)]),
    "rdf_eval_forest_stats": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int,
                                       _c_void_p, _c_int, _c_void_p, _c_int, _c_float, _c_void_p, _c_void_p]),
    "rdf_mean_shift_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "rdf_mean_shift": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_fingertip_heights": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int,
                                       _c_float, _c_float, _c_float, _c_float, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_mean_shift_heights": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int,
                                        _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_void_p,
                                        _c_void_p, _c_void_p]),
    "rdf_convert_0s_to_maxuint": (_c_int, [_c_void_p, _c_size_t, _c_void_p]),
    "rdf_setup_depth_image_for_forest": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "rdf_stencil_depth_image_by_group": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_flip_x": (_c_int, [_c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_prepare_hand_depth": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "rdf_make_rgba_from_labels": (_c_int, [_c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_init": (_c_int, [_c_void_p, _c_size_t, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_histogram": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int,
                                     _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "rdf_train_histogram_left": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int,
                                          _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "rdf_train_histogram_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "rdf_train_histogram_left_ws": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int,
                                             _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_sort_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "rdf_train_bits_row_bytes": (_c_size_t, [_c_int]),
    "rdf_train_sort_pixels": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_bits_workspace_bytes": (_c_size_t, [_c_int]),
    "rdf_train_decision_bits": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_count_rows": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                                      _c_void_p]),
    "rdf_train_right_counts": (_c_int, [_c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                        _c_void_p]),
    "rdf_train_pick_best": (_c_int, [_c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                     _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_next_active": (_c_int, [_c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "rdf_train_update_pixels": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                         _c_void_p]),
    "rdf_fill_u16": (_c_int, [_c_void_p, _c_size_t, ctypes.c_uint16, _c_void_p]),
    "rdf_stream_create_with_reserved_cus": (_c_int, [_c_void_p, _c_int]),
    "rdf_stream_destroy": (_c_int, [_c_void_p]),
    "rdf_stream_capture_id": (_c_int, [_c_void_p, ctypes.POINTER(ctypes.c_uint64)]),
    "rdf_graph_slots_release": (_c_int, [ctypes.c_uint64]),
    "rdf_debug_sched_slots": (_c_int, [_c_void_p, _c_void_p]),
    "rdf_debug_host_overhead": (_c_int, [_c_void_p, _c_int]),
    "rdf_device_malloc": (_c_int, [_c_void_p, _c_size_t]),
    "rdf_device_free": (_c_int, [_c_void_p]),
    "rdf_ipc_export": (_c_int, [_c_void_p, _c_void_p]),
    "rdf_ipc_open": (_c_int, [_c_void_p, _c_void_p]),
    "rdf_ipc_close": (_c_int, [_c_void_p]),
    "rdf_memcpy_device_async": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "rdf_debug_fat_kernel": (_c_int, [_c_int, ctypes.c_uint64, _c_void_p, _c_void_p]),
    "rdf_debug_floor_i32": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "rdf_debug_div_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "rdf_set_lds_budget_bytes": (None, [_c_int]),
    "rdf_set_block_threads": (None, [_c_int]),
    "rdf_set_scheduler": (None, [_c_int]),
    "rdf_set_compaction": (None, [_c_int]),
    "rdf_set_halo": (None, [_c_int]),
    "rdf_set_lds_levels": (None, [_c_int]),
    "rdf_set_tree_waves": (None, [_c_int]),
    "rdf_set_stage_vec": (None, [_c_int]),
    "rdf_set_group": (None, [_c_int]),
    "rdf_set_layers_one_launch": (None, [_c_int]),
    "rdf_set_rows_per_wave": (None, [_c_int]),
    "rdf_set_force_exact": (None, [_c_int]),
    "rdf_set_last_level_table": (None, [_c_int]),
    "rdf_set_deep_from": (None, [_c_int]),
    "rdf_set_fold": (None, [_c_int]),
    "rdf_event_create": (_c_int, [ctypes.POINTER(_c_void_p)]),
    "rdf_event_record": (_c_int, [_c_void_p, _c_void_p]),
    "rdf_event_synchronize": (_c_int, [_c_void_p]),
    "rdf_event_elapsed_ms": (_c_int, [_c_void_p, _c_void_p, ctypes.POINTER(_c_float)]),
    "rdf_event_destroy": (_c_int, [_c_void_p]),
    "rdf_stream_synchronize": (_c_int, [_c_void_p]),
    "rdf_abi_version": (_c_int, []),
    "rdf_build_id": (ctypes.c_char_p, []),
    "rdf_error_string": (ctypes.c_char_p, [_c_int]),
}

ABI_VERSION = 5
_lib = None


class RdfError(RuntimeError):
    pass


def library_path():
    # RDF_HIP_LIBRARY: alternate build of the same ABI (kernel timing experiments); default in-tree .so
    return os.environ.get("RDF_HIP_LIBRARY") or _build.SO


def load():
    """Open librdf_hip.so and type every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RdfError(f"{path} is missing: build it first (python __graft_entry__.py build, "
                       "or python 3d-beats_amd/_build.py). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.rdf_abi_version() != ABI_VERSION:
        raise RdfError(f"librdf_hip.so ABI {lib.rdf_abi_version()} != expected {ABI_VERSION}; rebuild")
    check_build_id(lib, path)
    _lib = lib
    return lib


def check_build_id(lib, path):
    """The library must have been built from the sources that sit next to it: same ABI number, yesterday's kernels would
    otherwise pass every check.  RDF_HIP_LIBRARY (an alternate build for a timing experiment) and a deployment without
    sources are exempt; RDF_ALLOW_STALE_LIBRARY=1 turns the refusal into a warning."""
    got = lib.rdf_build_id()
    got = got.decode() if isinstance(got, bytes) else str(got)
    if os.environ.get("RDF_HIP_LIBRARY") or not _build.sources_present():
        return got
    want = _build.source_id()
    if got != want:
        msg = (f"{path} was built from other sources (build id {got}, sources {want}): rebuild it "
               "(python __graft_entry__.py build).")
        if os.environ.get("RDF_ALLOW_STALE_LIBRARY") == "1":
            import warnings
            warnings.warn(msg)
        else:
            raise RdfError(msg)
    return got


def check(lib, code, what):
    if code != 0:
        msg = lib.rdf_error_string(int(code))
        msg = msg.decode() if isinstance(msg, bytes) else str(msg)
        raise RdfError(f"{what} failed: {msg} (code {code})")
