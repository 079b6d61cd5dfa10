"""ctypes binding of libkanter_core_amd.so (the C ABI in include/kanter_core_amd.h).

The shared library is the product; this module only declares its signatures.  If the library
has not been built the import fails loudly -- there is no Python or CPU fallback path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkanter_core_amd.so")

c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
c_fp = C.POINTER(C.c_float)
c_vp = C.c_void_p


class kc_size(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class kc_edge(C.Structure):
    _fields_ = [("output_id", C.c_uint32), ("input_id", C.c_uint32), ("output_slot", C.c_uint32),
                ("input_slot", C.c_uint32)]


class kc_node_desc(C.Structure):
    _fields_ = [("node_id", C.c_uint32), ("node_type", C.c_int32), ("mix_type", C.c_int32), ("value", C.c_float),
                ("embed_id", C.c_uint32), ("text", C.c_char_p), ("graph", c_vp), ("resize_policy", C.c_int32),
                ("policy_slot", C.c_uint32), ("policy_size", kc_size), ("resize_filter", C.c_int32)]


class kc_placement(C.Structure):
    _fields_ = [("node_id", C.c_uint32), ("rank", C.c_int32), ("component", C.c_int32), ("kind", C.c_int32)]


class kc_transfer(C.Structure):
    _fields_ = [("node_id", C.c_uint32), ("slot_id", C.c_uint32), ("src_rank", C.c_int32), ("dst_rank", C.c_int32),
                ("level", C.c_int32)]


class kc_band_range(C.Structure):
    _fields_ = [("y0", C.c_int32), ("y1", C.c_int32)]


class kc_band_rows(C.Structure):
    _fields_ = [("node_id", C.c_uint32), ("y0", C.c_int32), ("y1", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32)]


# name -> (restype, argtypes); every symbol include/kanter_core_amd.h declares.
SIGNATURES = {
    "kc_init": (C.c_int, [C.c_int]),
    "kc_shutdown": (C.c_int, []),
    "kc_is_initialized": (C.c_int, []),
    "kc_set_stream": (C.c_int, [c_vp]),
    "kc_get_stream": (c_vp, []),
    "kc_sync": (C.c_int, []),
    "kc_last_error": (C.c_char_p, []),
    "kc_status_string": (C.c_char_p, [C.c_int]),
    "kc_set_fusion": (C.c_int, [C.c_int]),
    "kc_set_resize_mode": (C.c_int, [C.c_int]),
    "kc_get_resize_mode": (C.c_int, []),
    "kc_set_cache_policy": (C.c_int, [C.c_int]),
    "kc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kc_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "kc_comm_destroy": (C.c_int, []),
    "kc_comm_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kc_comm_stats": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kc_live_graph_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "kc_comm_transport": (C.c_int, [C.c_char_p, C.c_size_t]),
    "kc_u8_pipe_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kc_u8_pipe_free": (C.c_int, [C.c_void_p]),
    "kc_u8_pipe_buffers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "kc_u8_pipe_upload": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "kc_u8_pipe_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "kc_u8_pipe_wait_download": (C.c_int, [C.c_void_p, C.c_int]),
    "kc_kernel_cache_set_dir": (C.c_int, [C.c_char_p]),
    "kc_kernel_cache_stats": (C.c_int, [C.POINTER(C.c_uint64)] * 4),
    "kc_kernel_cache_precompile": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p]),
    "kc_specialize_reset": (C.c_int, []),
    "kc_comm_gather_bands": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "kc_live_graph_evaluate_partitioned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "kc_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "kc_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "kc_get_cache_policy": (C.c_int, []),
    "kc_stats_counter": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64)]),
    "kc_resize_down2_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                       C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_float), C.c_size_t]),
    "kc_resize_upsample_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_float), C.c_size_t]),
    "kc_get_fusion": (C.c_int, []),
    "kc_stats": (C.c_int, [C.POINTER(C.c_uint64)] * 3),
    "kc_set_specialize": (C.c_int, [C.c_int, C.c_int]),
    "kc_get_specialize": (C.c_int, []),
    "kc_specialize_wait": (C.c_int, []),
    "kc_specialize_stats": (C.c_int, [C.POINTER(C.c_uint64)] * 4),
    "kc_specialize_compile_check": (C.c_int, [c_u32p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "kc_specialize_compile_check_upsample": (C.c_int, [c_u32p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_char_p,
                                                       C.c_size_t]),
    "kc_stats_algorithmic_bytes": (C.c_int, [C.POINTER(C.c_uint64)]),
    "kc_pool_trim": (C.c_int, []),
    "kc_live_graph_partition": (C.c_int, [c_vp, C.c_uint32, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "kc_partition_free": (C.c_int, [c_vp]),
    "kc_partition_info": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kc_partition_nodes": (C.c_int, [c_vp, C.POINTER(kc_placement), C.c_uint32, c_u32p]),
    "kc_partition_transfers": (C.c_int, [c_vp, C.POINTER(kc_transfer), C.c_uint32, c_u32p]),
    "kc_partition_kind": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "kc_partition_bands": (C.c_int, [c_vp, C.POINTER(kc_band_range), C.c_uint32, c_u32p, c_u32p, c_u32p]),
    "kc_partition_set_gather": (C.c_int, [c_vp, C.c_int]),
    "kc_live_graph_import_slot_data": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, c_vp]),
    "kc_live_graph_evaluate_band": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(c_vp)]),
    "kc_live_graph_band_source_rows": (C.c_int, [c_vp, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(kc_band_rows), C.c_uint32, c_u32p]),
    "kc_live_graph_embed_slot_data_band": (C.c_int, [c_vp, c_vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32]),
    "kc_plane_alloc": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(c_vp)]),
    "kc_plane_const": (C.c_int, [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(c_vp)]),
    "kc_plane_wrap": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.c_size_t, C.POINTER(c_vp)]),
    "kc_plane_retain": (C.c_int, [c_vp]),
    "kc_plane_release": (C.c_int, [c_vp]),
    "kc_plane_size": (C.c_int, [c_vp, c_u32p, c_u32p]),
    "kc_plane_is_const": (C.c_int, [c_vp, C.POINTER(C.c_int), c_fp]),
    "kc_plane_materialize": (C.c_int, [c_vp]),
    "kc_plane_device_ptr": (C.c_int, [c_vp, C.POINTER(c_vp), C.POINTER(C.c_size_t)]),
    "kc_plane_upload_f32": (C.c_int, [c_vp, c_vp, C.c_size_t]),
    "kc_plane_download_f32": (C.c_int, [c_vp, c_vp, C.c_size_t]),
    "kc_image_gray": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_image_rgba": (C.c_int, [C.POINTER(c_vp), C.POINTER(c_vp)]),
    "kc_image_retain": (C.c_int, [c_vp]),
    "kc_image_release": (C.c_int, [c_vp]),
    "kc_image_is_rgba": (C.c_int, [c_vp, C.POINTER(C.c_int)]),
    "kc_image_size": (C.c_int, [c_vp, C.POINTER(kc_size)]),
    "kc_image_plane": (C.c_int, [c_vp, C.c_int, C.POINTER(c_vp)]),
    "kc_image_from_value": (C.c_int, [kc_size, C.c_float, C.c_int, C.POINTER(c_vp)]),
    "kc_image_as_type": (C.c_int, [c_vp, C.c_int, C.POINTER(c_vp)]),
    "kc_image_materialize": (C.c_int, [c_vp]),
    "kc_image_from_u8": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(c_vp)]),
    "kc_image_to_u8": (C.c_int, [c_vp, C.c_int, c_vp]),
    "kc_image_from_f32": (C.c_int, [C.POINTER(c_vp), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(c_vp)]),
    "kc_image_to_f32": (C.c_int, [c_vp, C.POINTER(c_vp), C.c_int]),
    "kc_image_read_png": (C.c_int, [C.c_char_p, C.POINTER(c_vp)]),
    "kc_image_write_png": (C.c_int, [c_vp, C.c_char_p]),
    "kc_calculate_size": (C.c_int, [C.c_int, C.POINTER(kc_size), C.c_int, C.c_int, kc_size, C.POINTER(kc_size)]),
    "kc_resize_image": (C.c_int, [c_vp, kc_size, C.c_int, C.POINTER(c_vp)]),
    "kc_resize_buffers": (C.c_int, [C.POINTER(c_vp), C.POINTER(kc_edge), C.c_int, C.POINTER(kc_edge), C.c_int, C.c_int,
                                    C.c_uint32, kc_size, C.c_int, C.POINTER(c_vp)]),
    "kc_mix_process": (C.c_int, [c_vp, c_vp, C.c_int, C.POINTER(c_vp)]),
    "kc_separate_rgba_process": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_combine_rgba_process": (C.c_int, [C.POINTER(c_vp), C.POINTER(c_vp)]),
    "kc_value_process": (C.c_int, [C.c_float, C.POINTER(c_vp)]),
    "kc_height_to_normal_process": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_node_graph_new": (C.c_int, [C.POINTER(c_vp)]),
    "kc_node_graph_clone": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_node_graph_free": (C.c_int, [c_vp]),
    "kc_node_graph_from_path": (C.c_int, [C.c_char_p, C.POINTER(c_vp)]),
    "kc_node_graph_from_json": (C.c_int, [C.c_char_p, C.POINTER(c_vp)]),
    "kc_node_graph_export_json": (C.c_int, [c_vp, C.c_char_p]),
    "kc_node_graph_to_json": (C.c_int, [c_vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kc_node_graph_add_node": (C.c_int, [c_vp, C.POINTER(kc_node_desc), c_u32p]),
    "kc_node_graph_add_node_with_id": (C.c_int, [c_vp, C.POINTER(kc_node_desc)]),
    "kc_node_graph_connect": (C.c_int, [c_vp] + [C.c_uint32] * 4),
    "kc_node_graph_try_connect": (C.c_int, [c_vp] + [C.c_uint32] * 4),
    "kc_node_graph_remove_node": (C.c_int, [c_vp, C.c_uint32]),
    "kc_node_graph_remove_edge": (C.c_int, [c_vp, kc_edge]),
    "kc_node_graph_disconnect_slot": (C.c_int, [c_vp, C.c_uint32, C.c_int, C.c_uint32]),
    "kc_node_graph_node_count": (C.c_int, [c_vp, c_u32p]),
    "kc_node_graph_node_ids": (C.c_int, [c_vp, c_u32p, C.c_uint32, c_u32p]),
    "kc_node_graph_edges": (C.c_int, [c_vp, C.POINTER(kc_edge), C.c_uint32, c_u32p]),
    "kc_node_graph_input_slot_id_with_name": (C.c_int, [c_vp, C.c_char_p, c_u32p]),
    "kc_node_graph_output_slot_id_with_name": (C.c_int, [c_vp, C.c_char_p, c_u32p]),
    "kc_node_graph_set_mix_type": (C.c_int, [c_vp, C.c_uint32, C.c_int]),
    "kc_node_graph_set_image_node_path": (C.c_int, [c_vp, C.c_uint32, C.c_char_p]),
    "kc_node_graph_rename_output_node": (C.c_int, [c_vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]),
    "kc_tex_pro_new": (C.c_int, [C.c_uint64, C.POINTER(c_vp)]),
    "kc_tex_pro_free": (C.c_int, [c_vp]),
    "kc_tex_pro_new_live_graph": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_live_graph_free": (C.c_int, [c_vp]),
    "kc_live_graph_set_flags": (C.c_int, [c_vp, C.c_int, C.c_int]),
    "kc_live_graph_get_flags": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kc_live_graph_set_node_graph": (C.c_int, [c_vp, c_vp]),
    "kc_live_graph_node_graph": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "kc_live_graph_add_node": (C.c_int, [c_vp, C.POINTER(kc_node_desc), c_u32p]),
    "kc_live_graph_add_node_with_id": (C.c_int, [c_vp, C.POINTER(kc_node_desc)]),
    "kc_live_graph_remove_node": (C.c_int, [c_vp, C.c_uint32]),
    "kc_live_graph_connect": (C.c_int, [c_vp] + [C.c_uint32] * 4),
    "kc_live_graph_remove_edge": (C.c_int, [c_vp, kc_edge]),
    "kc_live_graph_disconnect_slot": (C.c_int, [c_vp, C.c_uint32, C.c_int, C.c_uint32]),
    "kc_live_graph_set_mix_type": (C.c_int, [c_vp, C.c_uint32, C.c_int]),
    "kc_live_graph_rename_output_node": (C.c_int, [c_vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]),
    "kc_live_graph_set_resize": (C.c_int, [c_vp, C.c_uint32, C.c_int, C.c_uint32, kc_size, C.c_int]),
    "kc_live_graph_node_state": (C.c_int, [c_vp, C.c_uint32, C.POINTER(C.c_int)]),
    "kc_live_graph_request": (C.c_int, [c_vp, C.c_uint32]),
    "kc_live_graph_prioritise": (C.c_int, [c_vp, C.c_uint32]),
    "kc_live_graph_await_clean": (C.c_int, [c_vp, C.c_uint32]),
    "kc_live_graph_update": (C.c_int, [c_vp]),
    "kc_live_graph_slot_data": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.POINTER(c_vp)]),
    "kc_live_graph_slot_data_size": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.POINTER(kc_size)]),
    "kc_live_graph_slot_in_memory": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]),
    "kc_live_graph_node_slot_ids": (C.c_int, [c_vp, C.c_uint32, c_u32p, C.c_uint32, c_u32p]),
    "kc_live_graph_buffer_rgba": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, C.c_int, c_vp]),
    "kc_live_graph_embed_slot_data_with_id": (C.c_int, [c_vp, c_vp, C.c_uint32, C.c_uint32]),
    "kc_live_graph_add_input_slot_data": (C.c_int, [c_vp, C.c_uint32, C.c_uint32, c_vp]),
    "kc_live_graph_changed_consume": (C.c_int, [c_vp, c_u32p, C.c_uint32, c_u32p]),
    "kc_live_graph_output_ids": (C.c_int, [c_vp, c_u32p, C.c_uint32, c_u32p]),
    "kc_live_graph_node_ids": (C.c_int, [c_vp, c_u32p, C.c_uint32, c_u32p]),
    "kc_live_graph_edges": (C.c_int, [c_vp, C.POINTER(kc_edge), C.c_uint32, c_u32p]),
    "kc_live_graph_set_base_dir": (C.c_int, [c_vp, C.c_char_p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7, same as
    /opt/rocm's).  A process must hold exactly ONE HIP runtime, otherwise whichever is loaded
    second finds no GPU and streams / device pointers cannot be shared.  Loading torch's copy
    first (by path, without importing torch) makes the loader resolve our DT_NEEDED
    libamdhip64.so.7 to it, and torch later re-opens the same file."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # already loaded: our library binds to it by soname
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Loads the HIP library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        # KC_LIB_PATH: a tuning build of the same library (tools/build_variant.sh) instead of the shipped one -- A/B runs point the
        # loader at the variant, nothing is copied over the product
        path = os.environ.get("KC_LIB_PATH") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                "kanter_core_amd: %s is missing. Build it with `python -m kanter_core_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU / pure-Python fallback." % path)
        _share_hip_runtime_with_torch()
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        # The kernel specialiser's compile thread is stopped while the interpreter is still whole (the library
        # also stops it from a C exit handler; this one runs first and keeps pending compiles from delaying exit).
        import atexit
        atexit.register(lambda: lib.kc_set_specialize(0, 0))
    return _lib
