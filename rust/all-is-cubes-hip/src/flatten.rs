//! Flattens a [`Space`] into the arrays of `aic_space_desc`: exactly the data `SpaceRaytracer::new` snapshots
//! (all-is-cubes-render/src/raytracer/sr.rs:64-88 and `prepare_cubes`, 543-549), in the layout of
//! `include/aic_hip.h`.

use all_is_cubes::block::{AIR, Evoxel, Evoxels};
use all_is_cubes::math::{Cube, Face};
use all_is_cubes::space::{self, Sky, SpaceBlockData};

use crate::ffi;

/// Owns the arrays an `aic_space_desc` points into.
pub(crate) struct FlatSpace {
    lo: [i32; 3],
    size: [i32; 3],
    block_index: Vec<u16>,
    light: Vec<u8>,
    blocks: Vec<ffi::aic_block_desc>,
    voxels: Vec<u16>,
    palette: Vec<f32>,
    sky_kind: i32,
    sky: [[f32; 3]; 8],
    block_sky: [[u8; 4]; 7],
}

/// One block-table entry with its own (not yet pooled) voxel and palette data: what `aic_replace_block` takes.
pub(crate) struct FlatBlock {
    pub desc: ffi::aic_block_desc,
    pub voxels: Vec<u16>,
    pub palette: Vec<f32>,
}

fn push_evoxel(out: &mut Vec<f32>, v: &Evoxel) {
    let c = v.color;
    out.extend_from_slice(&[
        c.red().into_inner(),
        c.green().into_inner(),
        c.blue().into_inner(),
        c.alpha().into_inner(),
        v.emission.red().into_inner(),
        v.emission.green().into_inner(),
        v.emission.blue().into_inner(),
        0.0,
    ]);
}

/// `TracingBlock::from_block` (sr.rs:578-587): the block's `Evoxels`, as palette indices + palette.
pub(crate) fn flatten_block(sbd: &SpaceBlockData) -> FlatBlock {
    let voxels: &Evoxels = sbd.evaluated().voxels();
    let is_air = sbd.block() == &AIR; // TracingCubeData.always_invisible (sr.rs:547)
    let mut flags = if is_air { ffi::AIC_BLOCK_AIR } else { 0 };
    if let Some(single) = voxels.single_voxel().filter(|_| voxels.resolution() == all_is_cubes::block::Resolution::R1) {
        // Evoxels::One, or a resolution-1 volume (voxel_storage.rs:364-385)
        flags |= ffi::AIC_BLOCK_ONE;
        let mut palette = Vec::with_capacity(8);
        push_evoxel(&mut palette, &single);
        return FlatBlock {
            desc: ffi::aic_block_desc { resolution: 1, vlo: [0; 3], vsize: [1; 3], pal_len: 1, flags, ..Default::default() },
            voxels: vec![0],
            palette,
        };
    }
    // paletted volume: `indices()` is already Z-major within `bounds()` (vol.rs:988-1023)
    let bounds = voxels.bounds();
    let lower = bounds.lower_bounds();
    let size = bounds.size();
    let mut palette = Vec::with_capacity(voxels.palette().len() * 8);
    for v in voxels.palette() {
        push_evoxel(&mut palette, v);
    }
    FlatBlock {
        desc: ffi::aic_block_desc {
            resolution: i32::from(voxels.resolution()),
            vlo: [lower.x, lower.y, lower.z],
            vsize: [size.width as i32, size.height as i32, size.depth as i32],
            pal_len: voxels.palette().len() as u32,
            flags,
            ..Default::default()
        },
        voxels: voxels.indices().as_linear().iter().map(|&i| u16::from(i)).collect(),
        palette,
    }
}

impl FlatSpace {
    pub(crate) fn new(space: &space::Read<'_>) -> Self {
        let bounds = space.bounds();
        let lower = bounds.lower_bounds();
        let size = bounds.size();
        let n = bounds.volume().expect("space volume fits usize");
        let mut block_index = Vec::with_capacity(n);
        let mut light = Vec::with_capacity(n * 4);
        // `Space::extract` visits cubes in the Vol's Z-major order, the order the device indexes (aic_device.h)
        let _ = space.extract(bounds, |e| {
            block_index.push(e.block_index());
            light.extend_from_slice(&e.light().as_texel()); // light/data.rs:160-170
        });

        let mut blocks = Vec::new();
        let mut voxels = Vec::new();
        let mut palette = Vec::new();
        for sbd in space.block_data() {
            let mut b = flatten_block(sbd);
            b.desc.vox_off = voxels.len() as u32;
            b.desc.pal_off = (palette.len() / 8) as u32;
            voxels.extend_from_slice(&b.voxels);
            palette.extend_from_slice(&b.palette);
            blocks.push(b.desc);
        }

        let sky_value = space.physics().sky.clone();
        let (sky_kind, sky) = match &sky_value {
            Sky::Uniform(c) => {
                let mut s = [[0.0; 3]; 8];
                s[0] = [c.red().into_inner(), c.green().into_inner(), c.blue().into_inner()];
                (0, s)
            }
            Sky::Octants(o) => (1, o.map(|c| [c.red().into_inner(), c.green().into_inner(), c.blue().into_inner()])),
            // Sky is #[non_exhaustive]: sample the octant directions of whatever it is
            other => (1, core::array::from_fn(|i| {
                let d = all_is_cubes::math::FreeVector::new(
                    if i & 4 != 0 { 1. } else { -1. }, if i & 2 != 0 { 1. } else { -1. }, if i & 1 != 0 { 1. } else { -1. });
                let c = other.sample(d);
                [c.red().into_inner(), c.green().into_inner(), c.blue().into_inner()]
            })),
        };
        let bs = sky_value.for_blocks(); // sky.rs:45-82
        let mut block_sky = [[0u8; 4]; 7];
        for (i, face) in Face::ALL.into_iter().enumerate() {
            block_sky[i] = bs.in_direction(face).as_texel();
        }
        block_sky[6] = bs.mean().as_texel();

        Self {
            lo: [lower.x, lower.y, lower.z],
            size: [size.width as i32, size.height as i32, size.depth as i32],
            block_index, light, blocks, voxels, palette, sky_kind, sky, block_sky,
        }
    }

    /// The descriptor; valid while `self` is alive and unmoved.
    pub(crate) fn desc(&self) -> ffi::aic_space_desc {
        ffi::aic_space_desc {
            lo: self.lo,
            size: self.size,
            block_index: self.block_index.as_ptr(),
            light: self.light.as_ptr(),
            n_blocks: self.blocks.len() as u32,
            blocks: self.blocks.as_ptr(),
            voxels: self.voxels.as_ptr(),
            n_voxels: self.voxels.len() as u64,
            palette: self.palette.as_ptr(),
            n_palette: (self.palette.len() / 8) as u64,
            sky_kind: self.sky_kind,
            sky: self.sky,
            block_sky: self.block_sky,
        }
    }
}

/// The per-cube part of an incremental update (updating.rs:155-166).
pub(crate) fn gather_cubes(space: &space::Read<'_>, cubes: impl Iterator<Item = Cube>) -> (Vec<i32>, Vec<u16>, Vec<u8>) {
    let (mut xyz, mut idx, mut light) = (Vec::new(), Vec::new(), Vec::new());
    for cube in cubes {
        xyz.extend_from_slice(&[cube.x, cube.y, cube.z]);
        idx.push(space.get_block_index(cube).unwrap_or(0));
        light.extend_from_slice(&space.get_light(cube).as_texel());
    }
    (xyz, idx, light)
}
