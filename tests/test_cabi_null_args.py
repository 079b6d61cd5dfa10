"""Every C-ABI entry point that takes a pointer is called with all-zero arguments (NULL handles, NULL outputs,
zero sizes): it must come back with a status, not crash.  No GPU needed.  Also part of tools/sanitize_host.sh."""
import ctypes as C

import pytest

from kanter_core_amd import _lib

POINTERISH = (C.c_void_p, C.c_char_p)
# NULL is an accepted argument of these: releasing / freeing nothing, switching back to the library's own stream,
# and the mix / combine operators, whose NULL images mean "input not connected" (src/node/mix.rs:57-83)
NULL_OK = {"kc_plane_release", "kc_image_release", "kc_node_graph_free", "kc_tex_pro_free", "kc_live_graph_free", "kc_partition_free",
           "kc_specialize_stats",  # every output is optional
           "kc_set_stream", "kc_stats", "kc_comm_info", "kc_comm_stats",  # kc_stats, kc_comm_info / _stats: every output is optional
           "kc_resize_buffers",  # n == 0: nothing to resize (src/shared.rs:147-149)
           "kc_kernel_cache_set_dir", "kc_kernel_cache_stats",  # NULL directory = the environment's choice; every output is optional
           "kc_u8_pipe_free"}  # freeing nothing


def _is_pointer(t):
    return t in POINTERISH or (isinstance(t, type) and issubclass(t, C._Pointer))


def _zero(t):
    if _is_pointer(t):
        return None
    return t()


CASES = sorted(name for name, (res, args) in _lib.SIGNATURES.items() if res is C.c_int and any(_is_pointer(a) for a in args))


@pytest.mark.parametrize("name", CASES)
def test_null_arguments_are_reported_not_dereferenced(name):
    L = _lib.load()
    res, args = _lib.SIGNATURES[name]
    status = getattr(L, name)(*[_zero(a) for a in args])
    if name in NULL_OK:
        return
    assert status != 0, "%s accepted NULL arguments" % name
    assert L.kc_status_string(status)  # a known status code with a message


def test_the_list_is_not_empty():
    assert len(CASES) > 60
