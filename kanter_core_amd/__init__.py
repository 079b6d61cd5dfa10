"""kanter_core_amd -- MI355X (gfx950) per-pixel evaluation backend for kanter_core / vismut_core
graphs.  The product is libkanter_core_amd.so (C ABI: include/kanter_core_amd.h, hand-written HIP
kernels in csrc/kernels.hip); this package mirrors the reference's host API on top of it.
Build the library with `python -m kanter_core_amd.build`; importing `api` symbols that touch the
device fails loudly when it is missing -- there is no CPU fallback.
"""
from .api import (Edge, EmbeddedSlotDataId, LiveGraph, MixType, Node, NodeGraph, NodeId, NodeState, NodeType,  # noqa: F401
                  ResizeFilter, ResizePolicy, Side, Size, SlotData, SlotId, SlotImage, TexProError,
                  TextureProcessor, calculate_size, combine_rgba_process, get_stream, height_to_normal_process,
                  init, is_initialized, mix_process, resize_image, separate_rgba_process, set_fusion, set_stream,
                  shutdown, stats, sync, value_process, set_specialize, get_specialize, specialize_wait,
                  specialize_stats, specialize_compile_check, Partition, PartitionPolicy, NodeKind, set_resize_mode,
                  get_resize_mode, resize_upsample_plan, resize_down2_plan, stats_counter, specialize_compile_check_upsample, set_cache_policy, get_cache_policy, set_option, get_option, comm_unique_id, comm_init,
                  comm_destroy, comm_info, comm_stats, comm_transport, comm_gather_bands, PlanKind, pool_trim, kernel_cache_set_dir, kernel_cache_stats,
                  kernel_cache_precompile, specialize_reset, U8Pipe)

__all__ = [n for n in dir() if not n.startswith("_")]
