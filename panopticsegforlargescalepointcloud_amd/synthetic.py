"""Seeded synthetic scenes of the shapes BASELINE.json names: an "urban" generator (55 % ground
plane, 25 % vertical facades, 20 % compact objects), voxelised like GridSampling3D(quantize_coords=True, mode="last")
(one REAL point per voxel, integer coords = round(pos / voxel); torch_points3d/core/data_transform/grid_transform.py:181-198),
cut into overlapping vertical cylinders on a grid like GridCylinderSampling
(torch_points3d/core/data_transform/transforms.py:182-267), with features x = (x_rel, y_rel, z_rel, z)
(conf/data/panoptic/npm3d-sparseconv_grid_012_R_16_cylinder_area1.yaml:60-74).

Also synthesises head outputs of realistic statistics (semantic class of the generator, offset = centre - p + N(0, 5 cm),
embedding = e_instance + N(0, 0.15)) so the clustering stage can be exercised without trained weights.
NumPy only; no dataset, no network.
"""
import numpy as np

NPM3D_NUM_CLASSES = 9
NPM3D_STUFF = (0, 1, 5)  # torch_points3d/datasets/panoptic/npm3d.py:47-48
THING_CLASSES = (2, 3, 4, 6, 7, 8)


def urban_points(n_points, extent, rng, objects_per_m2=0.12):
    """Raw (un-voxelised) points with class and instance id. Returns pos [n,3] f32, cls [n] i64, inst [n] i64 (0=none)."""
    n_ground = int(0.55 * n_points)
    n_fac = int(0.25 * n_points)
    n_obj = n_points - n_ground - n_fac
    g = np.empty((n_ground, 3), np.float32)
    g[:, :2] = rng.uniform(0, extent, size=(n_ground, 2))
    g[:, 2] = rng.normal(0, 0.02, n_ground)
    n_walls = max(4, int(extent / 6))
    w_id = rng.integers(0, n_walls, n_fac)
    w_org = rng.uniform(0, extent, size=(n_walls, 2))
    w_dir = rng.uniform(0, np.pi, n_walls)
    w_len = rng.uniform(8, 25, n_walls)
    t = rng.uniform(0, 1, n_fac) * w_len[w_id]
    f = np.empty((n_fac, 3), np.float32)
    f[:, 0] = w_org[w_id, 0] + np.cos(w_dir[w_id]) * t + rng.normal(0, 0.01, n_fac)
    f[:, 1] = w_org[w_id, 1] + np.sin(w_dir[w_id]) * t + rng.normal(0, 0.01, n_fac)
    f[:, 2] = rng.uniform(0, 8, n_fac)
    n_inst = max(8, int(objects_per_m2 * extent * extent))
    o_id = rng.integers(0, n_inst, n_obj)
    o_cen = rng.uniform(0, extent, size=(n_inst, 2))
    o_cls = np.asarray(THING_CLASSES)[rng.integers(0, len(THING_CLASSES), n_inst)]
    o = np.empty((n_obj, 3), np.float32)
    o[:, :2] = o_cen[o_id] + rng.normal(0, 0.5, size=(n_obj, 2))
    o[:, 2] = np.abs(rng.normal(0, 1.5, n_obj))
    pos = np.concatenate([g, f, o])
    cls = np.concatenate([np.zeros(n_ground, np.int64), np.ones(n_fac, np.int64), o_cls[o_id]])
    inst = np.concatenate([np.zeros(n_ground + n_fac, np.int64), o_id + 1])
    inside = (pos[:, 0] >= 0) & (pos[:, 0] <= extent) & (pos[:, 1] >= 0) & (pos[:, 1] <= extent)  # facades may overshoot
    return pos[inside], cls[inside], inst[inside]


FOR_NUM_CLASSES = 2
FOR_STUFF = (0,)  # torch_points3d/datasets/panoptic/treeins.py:34-36 (non-tree = stuff, tree = thing)


def forest_points(n_points, extent, rng, trees_per_m2=0.04):
    """FOR-instance-like raw points: 60 % ground, 40 % trees (stem = vertical cylinder r 0.2 m,
    h ~15 m; crown = ellipsoid shell on top).  Returns pos, cls (0 ground / 1 tree), inst (0 = none)."""
    n_ground = int(0.6 * n_points)
    n_tree = n_points - n_ground
    g = np.empty((n_ground, 3), np.float32)
    g[:, :2] = rng.uniform(0, extent, size=(n_ground, 2))
    g[:, 2] = 0.3 * np.sin(g[:, 0] / 7.0) + rng.normal(0, 0.03, n_ground)
    n_inst = max(4, int(trees_per_m2 * extent * extent))
    t_id = rng.integers(0, n_inst, n_tree)
    cen = rng.uniform(0, extent, size=(n_inst, 2))
    height = rng.uniform(10, 18, n_inst)
    crown_r = rng.uniform(1.5, 3.0, n_inst)
    is_stem = rng.random(n_tree) < 0.3
    t = np.empty((n_tree, 3), np.float32)
    ang = rng.uniform(0, 2 * np.pi, n_tree)
    # stems
    t[:, 0] = cen[t_id, 0] + 0.2 * np.cos(ang)
    t[:, 1] = cen[t_id, 1] + 0.2 * np.sin(ang)
    t[:, 2] = rng.uniform(0, 1, n_tree) * height[t_id] * 0.7
    # crowns: points on an ellipsoid shell around (cx, cy, 0.75 h)
    u = rng.normal(size=(n_tree, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    cr = ~is_stem
    t[cr, 0] = cen[t_id[cr], 0] + u[cr, 0] * crown_r[t_id[cr]]
    t[cr, 1] = cen[t_id[cr], 1] + u[cr, 1] * crown_r[t_id[cr]]
    t[cr, 2] = 0.75 * height[t_id[cr]] + u[cr, 2] * 0.3 * height[t_id[cr]]
    pos = np.concatenate([g, t]).astype(np.float32)
    cls = np.concatenate([np.zeros(n_ground, np.int64), np.ones(n_tree, np.int64)])
    inst = np.concatenate([np.zeros(n_ground, np.int64), t_id + 1])
    inside = (pos[:, 0] >= 0) & (pos[:, 0] <= extent) & (pos[:, 1] >= 0) & (pos[:, 1] <= extent)
    return pos[inside], cls[inside], inst[inside]


def forest_scene(n_voxels_target, voxel=0.10, seed=2022, raw_per_voxel=1.5):
    rng = np.random.default_rng(seed)
    extent = float(np.sqrt(0.6 * n_voxels_target / (0.7 / (voxel * voxel))))
    pos, cls, inst = forest_points(int(n_voxels_target * raw_per_voxel), extent, rng)
    pos, coords, cls, inst = voxelise(pos, cls, inst, voxel, rng)
    return Scene(pos, coords, cls, inst, voxel, extent)


def voxelise(pos, cls, inst, voxel, rng):
    """GridSampling3D(mode='last'): shuffle, keep one real point per voxel."""
    perm = rng.permutation(len(pos))
    pos, cls, inst = pos[perm], cls[perm], inst[perm]
    c = np.round(pos / voxel).astype(np.int64)
    key = (c[:, 0] + (1 << 20)) * (1 << 42) + (c[:, 1] + (1 << 20)) * (1 << 21) + (c[:, 2] + (1 << 20))
    _, first = np.unique(key, return_index=True)
    first.sort()
    return pos[first], c[first].astype(np.int32), cls[first], inst[first]


class Scene:
    """Voxelised scene: pos f32 [U,3], coords i32 [U,3], cls, inst (0 = stuff / none), voxel size."""

    def __init__(self, pos, coords, cls, inst, voxel, extent):
        self.pos, self.coords, self.cls, self.inst, self.voxel, self.extent = pos, coords, cls, inst, voxel, extent
        n_inst = int(inst.max()) + 1
        cnt = np.bincount(inst, minlength=n_inst).astype(np.float64)
        cen = np.stack([np.bincount(inst, weights=pos[:, d], minlength=n_inst) for d in range(3)], 1)
        self.inst_center = (cen / np.maximum(cnt, 1)[:, None]).astype(np.float32)
        self.n_inst = n_inst


def urban_scene(n_voxels_target, voxel=0.05, seed=2022, raw_per_voxel=1.6, ground_density=None):
    """Scene with ~n_voxels_target occupied voxels. The extent follows from the ground density at this voxel size."""
    rng = np.random.default_rng(seed)
    if ground_density is None:
        ground_density = 0.75 / (voxel * voxel)  # occupied ground voxels per m^2 at this sampling rate
    extent = float(np.sqrt(0.55 * n_voxels_target / ground_density))
    pos, cls, inst = urban_points(int(n_voxels_target * raw_per_voxel), extent, rng)
    pos, coords, cls, inst = voxelise(pos, cls, inst, voxel, rng)
    return Scene(pos, coords, cls, inst, voxel, extent)


def cylinder_tiles(scene, grid, radius_factor=0.85):
    """grid x grid overlapping vertical cylinders (spacing s = extent/grid, radius = radius_factor * s >= s/sqrt(2)).
    Returns a list of index arrays (origin ids into the scene), ordered row-major like the reference's block order."""
    s = scene.extent / grid
    r = radius_factor * s
    tiles = []
    xy = scene.pos[:, :2]
    for iy in range(grid):
        for ix in range(grid):
            c = np.array([(ix + 0.5) * s, (iy + 0.5) * s], np.float32)
            d2 = ((xy - c) ** 2).sum(1)
            tiles.append(np.nonzero(d2 < r * r)[0])
    return tiles, r


def tile_batch(scene, tiles, tile_ids):
    """Collate tiles into one batch (Batch.from_data_list for SPARSE, torch_points3d/datasets/base_dataset.py:171-175).
    pos is centred per cylinder; coords are the scene's integer coords shifted by the rounded tile centre."""
    pos, coords, batch, x, origin = [], [], [], [], []
    for b, t in enumerate(tile_ids):
        idx = tiles[t]
        p = scene.pos[idx]
        cen = p.mean(0)
        cen[2] = 0.0
        pc = p - cen
        ci = scene.coords[idx] - np.round(cen / scene.voxel).astype(np.int32)
        rel = pc - pc.mean(0)
        pos.append(pc.astype(np.float32))
        coords.append(ci.astype(np.int32))
        batch.append(np.full(len(idx), b, np.int64))
        x.append(np.concatenate([rel, pc[:, 2:3]], 1).astype(np.float32))
        origin.append(idx)
    return {"pos": np.concatenate(pos), "coords": np.concatenate(coords), "batch": np.concatenate(batch),
            "x": np.concatenate(x), "origin_id": np.concatenate(origin)}


def synthetic_head_outputs(scene, origin_id, pos_centred_offset, rng, embed_dim=5, offset_sigma=0.05, embed_sigma=0.15):
    """Semantic argmax / offsets / embeddings with trained-network statistics for the given tile rows.
    pos_centred_offset: (scene pos - tile pos) per row, so offsets point to the instance centre in tile coordinates."""
    cls = scene.cls[origin_id]
    inst = scene.inst[origin_id]
    e_inst = np.random.default_rng(12345).normal(0, 3.0, size=(scene.n_inst, embed_dim)).astype(np.float32)
    target = scene.inst_center[inst] - pos_centred_offset
    off = (target - (scene.pos[origin_id] - pos_centred_offset)).astype(np.float32)
    off += rng.normal(0, offset_sigma, size=off.shape).astype(np.float32)
    off[inst == 0] = 0
    emb = e_inst[inst] + rng.normal(0, embed_sigma, size=(len(inst), embed_dim)).astype(np.float32)
    return cls.astype(np.int64), off, emb.astype(np.float32)
