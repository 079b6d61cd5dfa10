// Replacement bodies for the reference's per-node pixel functions, written against
// kanter_core_amd_sys.rs.  SOURCE ONLY: there is no Rust toolchain in this build environment, so
// this file has never been compiled; it shows exactly where the C ABI plugs into vismut_core 0.10.0
// (paths and line numbers refer to the reference checkout).  The public API of the crate
// (NodeGraph / LiveGraph / TextureProcessor / Node / SlotData) does not change.
//
// Drop-in points:
//   src/slot_image.rs   SlotImage keeps its two variants but wraps a device image handle
//   src/node/mix.rs:51            mix::process
//   src/node/separate_rgba.rs:38  separate_rgba::process
//   src/node/combine_rgba.rs:14   combine_rgba::process
//   src/node/value.rs:14          value::process
//   src/node/height_to_normal.rs:16  height_to_normal::process
//   src/shared.rs:16,141,218      deconstruct_image / resize_buffers / read_slot_image
//   src/slot_image.rs:28,146,172,212  from_value / to_u8 / to_u8_srgb / as_type
use crate::kanter_core_amd_sys::*;
use crate::{edge::Edge, error::{Result, TexProError}, node::{mix::MixType, Node, ResizeFilter, ResizePolicy},
            node_graph::SlotId, slot_data::{Size, SlotData}};
use std::{ffi::CStr, os::raw::c_void, ptr, sync::Arc};

/// Owned reference to a device image (kc_image is reference counted by the library).
pub struct DeviceImage(ptr::NonNull<KcImage>);
unsafe impl Send for DeviceImage {}
unsafe impl Sync for DeviceImage {}
impl Clone for DeviceImage {
    fn clone(&self) -> Self { unsafe { kc_image_retain(self.0.as_ptr()) }; DeviceImage(self.0) }
}
impl Drop for DeviceImage {
    fn drop(&mut self) { unsafe { kc_image_release(self.0.as_ptr()) }; }
}

/// src/slot_image.rs:15-19 -- same two variants, planes live in HBM.
#[derive(Clone)]
pub enum SlotImage { Gray(DeviceImage), Rgba(DeviceImage) }

fn check(status: i32) -> Result<()> {
    // 1..=19 are TexProError in declaration order (src/error.rs:5-27); the two variants that carry a payload are built
    // from kc_last_error() (the library keeps the text, not the foreign error object); >= 100 are the library's own.
    let text = || unsafe { CStr::from_ptr(kc_last_error()) }.to_string_lossy().into_owned();
    match status {
        0 => Ok(()),
        1 => Err(TexProError::Generic),
        2 => Err(TexProError::Canceled),
        3 => Err(TexProError::Image(image::ImageError::IoError(std::io::Error::new(std::io::ErrorKind::Other, text())))),
        4 => Err(TexProError::InvalidBufferCount),
        5 => Err(TexProError::InvalidNodeId),
        6 => Err(TexProError::InvalidNodeType),
        7 => Err(TexProError::InvalidSlotId),
        8 => Err(TexProError::InvalidSlotType),
        9 => Err(TexProError::InvalidEdge),
        10 => Err(TexProError::NoSlotData),
        11 => Err(TexProError::SlotOccupied),
        12 => Err(TexProError::SlotNotOccupied),
        13 => Err(TexProError::UnableToLock),
        14 => Err(TexProError::NodeProcessing),
        15 => Err(TexProError::PoisonError),
        16 => Err(TexProError::TryLockError),
        17 => Err(TexProError::NodeDirty),
        18 => Err(TexProError::Io(std::io::Error::new(std::io::ErrorKind::Other, text()))),
        19 => Err(TexProError::InvalidName),
        _ => {
            // KC_ERR_HIP / NO_DEVICE / INVALID_ARG / OUT_OF_MEMORY / UNSUPPORTED: no counterpart in the reference
            eprintln!("kanter_core_amd: status {}: {}", status, text());
            Err(TexProError::Generic)
        }
    }
}

fn wrap(raw: *mut KcImage) -> SlotImage {
    let img = DeviceImage(ptr::NonNull::new(raw).expect("null image"));
    let mut rgba = 0;
    unsafe { kc_image_is_rgba(raw, &mut rgba) };
    if rgba != 0 { SlotImage::Rgba(img) } else { SlotImage::Gray(img) }
}

impl SlotImage {
    fn raw(&self) -> *mut KcImage { match self { Self::Gray(i) | Self::Rgba(i) => i.0.as_ptr() } }
    pub fn is_rgba(&self) -> bool { matches!(self, Self::Rgba(_)) }

    /// src/slot_image.rs:116-121
    pub fn size(&self) -> Result<Size> {
        let mut s = KcSize { width: 0, height: 0 };
        check(unsafe { kc_image_size(self.raw(), &mut s) })?;
        Ok(Size::new(s.width, s.height))
    }
    /// src/slot_image.rs:28-64
    pub fn from_value(size: Size, value: f32, rgba: bool) -> Self {
        let mut out = ptr::null_mut();
        let s = KcSize { width: size.width, height: size.height };
        check(unsafe { kc_image_from_value(s, value, rgba as i32, &mut out) }).unwrap();
        wrap(out)
    }
    /// src/slot_image.rs:212-256
    pub fn as_type(&self, rgba: bool) -> Result<Self> {
        let mut out = ptr::null_mut();
        check(unsafe { kc_image_as_type(self.raw(), rgba as i32, &mut out) })?;
        Ok(wrap(out))
    }
    /// src/slot_image.rs:146-170 (srgb = true: :172-207)
    fn to_u8_impl(&self, srgb: bool) -> Result<Vec<u8>> {
        let s = self.size()?;
        let mut v = vec![0u8; s.pixel_count() * 4];
        check(unsafe { kc_image_to_u8(self.raw(), srgb as i32, v.as_mut_ptr()) })?;
        Ok(v)
    }
    pub fn to_u8(&self) -> Result<Vec<u8>> { self.to_u8_impl(false) }
    pub fn to_u8_srgb(&self) -> Result<Vec<u8>> { self.to_u8_impl(true) }
}

/// src/shared.rs:218-261 (decode with the `image` crate as before, then hand the u8 samples over).
pub fn read_slot_image<P: AsRef<std::path::Path>>(path: P) -> Result<SlotImage> {
    let image = ::image::open(path)?;
    let px = image.as_flat_samples_u8().unwrap().samples;
    let (w, h) = (image.width(), image.height());
    let channels = (px.len() / (w * h) as usize) as i32;
    let mut out = ptr::null_mut();
    check(unsafe { kc_image_from_u8(px.as_ptr(), w, h, channels, &mut out) })?;
    Ok(wrap(out))
}

fn slot(slot_datas: &[Arc<SlotData>], id: u32) -> *mut KcImage {
    slot_datas.iter().find(|sd| sd.slot_id == SlotId(id)).map_or(ptr::null_mut(), |sd| sd.image.raw())
}

/// src/node/mix.rs:51-134
pub(crate) fn mix_process(slot_datas: &[Arc<SlotData>], node: &Node, mix_type: MixType) -> Result<Vec<Arc<SlotData>>> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_mix_process(slot(slot_datas, 0), slot(slot_datas, 1), mix_type as i32, &mut out) })?;
    Ok(if out.is_null() { Vec::new() } else { vec![Arc::new(SlotData::new(node.node_id, SlotId(0), wrap(out)))] })
}

/// src/node/separate_rgba.rs:38-69
pub(crate) fn separate_rgba_process(slot_datas: &[Arc<SlotData>], node: &Node) -> Result<Vec<Arc<SlotData>>> {
    let input = slot_datas.get(0).map_or(ptr::null_mut(), |sd| sd.image.raw());
    let mut outs = [ptr::null_mut(); 4];
    check(unsafe { kc_separate_rgba_process(input, outs.as_mut_ptr()) })?;
    Ok(outs.iter().enumerate().map(|(i, o)| Arc::new(SlotData::new(node.node_id, SlotId(i as u32), wrap(*o)))).collect())
}

/// src/node/combine_rgba.rs:14-97
pub(crate) fn combine_rgba_process(slot_datas: &[Arc<SlotData>], node: &Node) -> Result<Vec<Arc<SlotData>>> {
    let ins = [slot(slot_datas, 0), slot(slot_datas, 1), slot(slot_datas, 2), slot(slot_datas, 3)];
    let mut out = ptr::null_mut();
    check(unsafe { kc_combine_rgba_process(ins.as_ptr(), &mut out) })?;
    Ok(vec![Arc::new(SlotData::new(node.node_id, SlotId(0), wrap(out)))])
}

/// src/node/value.rs:14-26
pub(crate) fn value_process(node: &Node, value: f32) -> Vec<Arc<SlotData>> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_value_process(value, &mut out) }).unwrap();
    vec![Arc::new(SlotData::new(node.node_id, SlotId(0), wrap(out)))]
}

/// src/node/height_to_normal.rs:16-77 (a 100 us kernel: the per-pixel cancel poll has no analogue)
pub(crate) fn height_to_normal_process(slot_datas: &[Arc<SlotData>], node: &Node) -> Result<Vec<Arc<SlotData>>> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_height_to_normal_process(slot(slot_datas, 0), &mut out) })?;
    Ok(if out.is_null() { Vec::new() } else { vec![Arc::new(SlotData::new(node.node_id, SlotId(0), wrap(out)))] })
}

/// src/shared.rs:141-216 -- `edges` already sorted by input_slot (src/node/node_type.rs:230-231).
pub(crate) fn resize_buffers(slot_datas: &[Arc<SlotData>], edges: &[Edge], policy: ResizePolicy, filter: ResizeFilter)
    -> Result<Vec<Arc<SlotData>>> {
    if slot_datas.is_empty() { return Ok(slot_datas.into()); }
    let images: Vec<*mut KcImage> = slot_datas.iter().map(|sd| sd.image.raw()).collect();
    let key = |n: u32, s: u32| KcEdge { output_id: n, input_id: 0, output_slot: s, input_slot: 0 };
    let keys: Vec<KcEdge> = slot_datas.iter().map(|sd| key(sd.node_id.0, sd.slot_id.0)).collect();
    let es: Vec<KcEdge> = edges.iter().map(|e| KcEdge { output_id: e.output_id.0, input_id: e.input_id.0,
                                                        output_slot: e.output_slot.0, input_slot: e.input_slot.0 }).collect();
    let (p, pslot, psize) = match policy {
        ResizePolicy::MostPixels => (0, 0, KcSize { width: 0, height: 0 }),
        ResizePolicy::LeastPixels => (1, 0, KcSize { width: 0, height: 0 }),
        ResizePolicy::LargestAxes => (2, 0, KcSize { width: 0, height: 0 }),
        ResizePolicy::SmallestAxes => (3, 0, KcSize { width: 0, height: 0 }),
        ResizePolicy::SpecificSlot(s) => (4, s.0, KcSize { width: 0, height: 0 }),
        ResizePolicy::SpecificSize(s) => (5, 0, KcSize { width: s.width, height: s.height }),
    };
    let mut outs = vec![ptr::null_mut(); images.len()];
    check(unsafe { kc_resize_buffers(images.as_ptr(), keys.as_ptr(), images.len() as i32, es.as_ptr(), es.len() as i32,
                                     p, pslot, psize, filter as i32, outs.as_mut_ptr()) })?;
    Ok(slot_datas.iter().zip(outs).map(|(sd, o)| Arc::new(SlotData::new(sd.node_id, sd.slot_id, wrap(o)))).collect())
}

// ------------------------------------------------------------------------------------------------
// Several GPUs (one process per GPU).  The reference has no such layer; these are the calls a Rust host adds
// around `LiveGraph`: join the communicator once, make a plan, evaluate it.  The library moves the data itself
// (csrc/comm.cpp); INTEGRATION.md section 6 shows the whole sequence.
// ------------------------------------------------------------------------------------------------

/// How the evaluation of one node is spread over the ranks (kc_live_graph_partition, csrc/partition.cpp): the same answer
/// on every rank, derived from the graph alone.  Owns the library's plan.
pub struct Plan {
    raw: *mut KcPartition,
    /// 0 = one GPU, 1 = branches (`transfers` cross rank boundaries), 2 = row bands (`bands`)
    pub kind: i32,
    pub nodes: Vec<KcPlacement>,
    pub transfers: Vec<KcTransfer>,
    /// rows [y0, y1) of the requested node per rank (band plans)
    pub bands: Vec<KcBandRange>,
}

impl Drop for Plan {
    fn drop(&mut self) { unsafe { kc_partition_free(self.raw); } }
}

/// rank 0: the identifier every rank passes to `comm_init` (hand it over by any channel the host has)
pub fn comm_unique_id() -> Result<[u8; 256]> {
    let mut id = [0u8; 256];
    check(unsafe { kc_comm_unique_id(id.as_mut_ptr() as *mut c_void) })?;
    Ok(id)
}

/// collective over all ranks of the node, after `kc_init`
pub fn comm_init(rank: i32, world: i32, id: &[u8; 256]) -> Result<()> {
    check(unsafe { kc_comm_init(rank, world, id.as_ptr() as *const c_void) })
}

pub fn comm_destroy() -> Result<()> { check(unsafe { kc_comm_destroy() }) }

/// policy: 0 = the cheapest of {one GPU, branches, row bands + gather}, 1 = branches, 2 = row bands
pub fn partition(lg: *mut KcLiveGraph, root: NodeId, world: i32, policy: i32) -> Result<Plan> {
    let mut raw = ptr::null_mut();
    check(unsafe { kc_live_graph_partition(lg, root.0, world, policy, &mut raw) })?;
    let mut plan = Plan { raw, kind: 0, nodes: Vec::new(), transfers: Vec::new(), bands: Vec::new() };
    check(unsafe { kc_partition_kind(raw, &mut plan.kind, ptr::null_mut(), ptr::null_mut(), ptr::null_mut()) })?;
    let (mut n, mut t, mut b) = (0u32, 0u32, 0u32);
    check(unsafe { kc_partition_nodes(raw, ptr::null_mut(), 0, &mut n) })?;
    check(unsafe { kc_partition_transfers(raw, ptr::null_mut(), 0, &mut t) })?;
    check(unsafe { kc_partition_bands(raw, ptr::null_mut(), 0, &mut b, ptr::null_mut(), ptr::null_mut()) })?;
    plan.nodes = vec![KcPlacement { node_id: 0, rank: 0, component: 0, kind: 0 }; n as usize];
    plan.transfers = vec![KcTransfer { node_id: 0, slot_id: 0, src_rank: 0, dst_rank: 0, level: 0 }; t as usize];
    plan.bands = vec![KcBandRange { y0: 0, y1: 0 }; b as usize];
    check(unsafe { kc_partition_nodes(raw, plan.nodes.as_mut_ptr(), n, &mut n) })?;
    check(unsafe { kc_partition_transfers(raw, plan.transfers.as_mut_ptr(), t, &mut t) })?;
    check(unsafe { kc_partition_bands(raw, plan.bands.as_mut_ptr(), b, &mut b, ptr::null_mut(), ptr::null_mut()) })?;
    Ok(plan)
}

/// The whole evaluation of a plan (kc_live_graph_evaluate_partitioned): the exchange of a branch plan and the root on the
/// home rank, or this rank's rows of a band plan and the gather.  `Some(image)` on the home rank, `None` elsewhere.
pub fn evaluate_partitioned(lg: *mut KcLiveGraph, plan: &Plan, root: NodeId) -> Result<Option<SlotImage>> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_live_graph_evaluate_partitioned(lg, plan.raw, root.0, &mut out) })?;
    Ok(if out.is_null() { None } else { Some(wrap(out)) })
}

/// Every rank passes its band; the assembled image on `home` (kc_comm_gather_bands).
pub fn gather_bands(band: &SlotImage, y0: i32, full_height: u32, home: i32) -> Result<Option<SlotImage>> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_comm_gather_bands(band.raw(), y0, full_height, home, &mut out) })?;
    Ok(if out.is_null() { None } else { Some(wrap(out)) })
}

/// Rows [y0, y1) of `node`'s slot, bit-identical to those rows of the whole-image evaluation
/// (kc_live_graph_evaluate_band, csrc/bands.cpp): every rank takes its own band of the result.
pub fn evaluate_band(lg: *mut KcLiveGraph, node: NodeId, slot: SlotId, y0: i32, y1: i32) -> Result<SlotImage> {
    let mut out = ptr::null_mut();
    check(unsafe { kc_live_graph_evaluate_band(lg, node.0, slot.0, y0, y1, &mut out) })?;
    Ok(wrap(out))
}
