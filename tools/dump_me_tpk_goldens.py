"""Pins the part of the oracle that could not be pinned in the build container: MinkowskiEngine's convolution (kernel
maps, the order of the 27 offsets inside `.kernel`, strided / transposed output coordinates) and torch-points-kernels'
region_grow.  Neither library is under /root/reference nor installable there, so `oracle/` restates their
published behaviour and DESIGN.md 9 says "parity unpinned" for these rows.

Run this ONCE on any machine that has the reference's real dependencies (MinkowskiEngine 0.5.x, torch-points-kernels
0.7.0, a CUDA GPU):

    python tools/dump_me_tpk_goldens.py            # writes tests/golden/me_tpk_cases.npz

and commit the file.  tests/test_oracle.py::test_oracle_matches_minkowskiengine_and_tpk_goldens (CPU) picks it up when
present and skips otherwise; the HIP path is compared bit for bit with the oracle on the same kinds of input by the GPU
tests, so pinning the oracle pins the product.
Only inputs and outputs are stored -- seeded inputs from tests/bruteforce.py, no library source.

What is dumped, per case:
  coords                int32 [N,4]   (batch, x, y, z), unique, random row order (bruteforce.surface_coords, seeded)
  conv_same_onehot_k    for every k < 27: output of MinkowskiConvolution(1, 1, kernel_size=3, stride=1) whose kernel is
                        one-hot at offset k, fed with feature = 1 + row index  ->  out[o] = 1 + nbr_k(o), i.e. the kernel
                        map AND the position of every offset inside ME's kernel tensor
  conv_stride2_*        the same through a stride-2 convolution (kernel_size 3 and 2): output coordinates + one-hot outputs
  convtr_stride2_*      the same through MinkowskiConvolutionTranspose back onto the input's coordinate map
  conv_random           a random [27, 16, 32] kernel on random features (float parity of a whole layer)
  region_grow_*         tpk.region_grow(pos, labels, batch, ignore_labels, radius, nsample, min_cluster_size) on jittered
                        instance blobs incl. the truncating regime nsample << neighbours: clusters as (points, offsets)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import MinkowskiEngine as ME
    import torch_points_kernels as tpk
    import bruteforce as bf
    dev = torch.device("cuda")
    out = {"versions": np.array([ME.__version__, getattr(tpk, "__version__", "?"), torch.__version__])}
    for case, (seed, n_batch, n, extent) in enumerate([(11, 2, 600, 20), (12, 1, 1500, 30), (13, 3, 250, 12)]):
        rng = np.random.default_rng(seed)
        coords = bf.surface_coords(rng, n_batch=n_batch, n=n, extent=extent)
        N = len(coords)
        tag = "c%d_" % case
        out[tag + "coords"] = coords
        feats = (1.0 + np.arange(N, dtype=np.float32))[:, None]

        def sparse(f):
            return ME.SparseTensor(features=torch.from_numpy(f).to(dev), coordinates=torch.from_numpy(coords).to(dev), device=dev)

        def onehot_outputs(make_conv, x, name, coord_key=None):
            res = []
            for k in range(make_conv().kernel.shape[0]):
                conv = make_conv().to(dev)
                with torch.no_grad():
                    conv.kernel.zero_()
                    conv.kernel[k, 0, 0] = 1.0
                y = conv(x) if coord_key is None else conv(x, coord_key)
                res.append(y.F[:, 0].detach().cpu().numpy())
                if k == 0:
                    out[tag + name + "_out_coords"] = y.C.cpu().numpy().astype(np.int32)
            out[tag + name + "_onehot"] = np.stack(res)          # [K, N_out]: 1 + input row of offset k, 0 = none

        x = sparse(feats)
        onehot_outputs(lambda: ME.MinkowskiConvolution(1, 1, kernel_size=3, stride=1, bias=False, dimension=3), x, "conv_same")
        for ks in (3, 2):
            onehot_outputs(lambda: ME.MinkowskiConvolution(1, 1, kernel_size=ks, stride=2, bias=False, dimension=3), x, "conv_stride2_k%d" % ks)
            down = ME.MinkowskiConvolution(1, 1, kernel_size=ks, stride=2, bias=False, dimension=3).to(dev)
            with torch.no_grad():
                down.kernel.fill_(1.0)
            h = down(x)
            hf = ME.SparseTensor(features=(1.0 + torch.arange(h.F.shape[0], dtype=torch.float32, device=dev))[:, None],
                                 coordinate_map_key=h.coordinate_map_key, coordinate_manager=h.coordinate_manager)
            out[tag + "convtr_stride2_k%d_in_coords" % ks] = h.C.cpu().numpy().astype(np.int32)
            onehot_outputs(lambda: ME.MinkowskiConvolutionTranspose(1, 1, kernel_size=ks, stride=2, bias=False, dimension=3), hf,
                           "convtr_stride2_k%d" % ks)
        # a whole random layer
        w = (rng.normal(size=(27, 16, 32)) * 0.1).astype(np.float32)
        f = rng.normal(size=(N, 16)).astype(np.float32)
        conv = ME.MinkowskiConvolution(16, 32, kernel_size=3, stride=1, bias=False, dimension=3).to(dev)
        with torch.no_grad():
            conv.kernel.copy_(torch.from_numpy(w))
        y = conv(sparse(f))
        out[tag + "conv_random_w"], out[tag + "conv_random_in"] = w, f
        out[tag + "conv_random_out"], out[tag + "conv_random_out_coords"] = y.F.detach().cpu().numpy(), y.C.cpu().numpy().astype(np.int32)
    # ---- region_grow
    for case, (seed, n_inst, pts, sigma, nsample, radius) in enumerate([(21, 12, 120, 0.05, 16, 0.09), (22, 8, 400, 0.02, 32, 0.06),
                                                                        (23, 6, 900, 0.01, 200, 0.075), (24, 10, 60, 0.08, 4, 0.12)]):
        rng = np.random.default_rng(seed)
        centres = rng.uniform(0, 3.0, size=(n_inst, 3))
        inst = rng.integers(0, n_inst, size=n_inst * pts)
        pos = (centres[inst] + rng.normal(size=(len(inst), 3)) * sigma).astype(np.float32)
        labels = (inst % 4).astype(np.int64)                        # class 0 is ignored below
        batch = (inst >= n_inst // 2).astype(np.int64)
        order = np.argsort(batch, kind="stable")
        pos, labels, batch = pos[order], labels[order], batch[order]
        clusters = tpk.region_grow(torch.from_numpy(pos).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(batch).to(dev),
                                   ignore_labels=torch.tensor([0]).to(dev), radius=radius, nsample=nsample, min_cluster_size=10)
        tag = "rg%d_" % case
        out[tag + "pos"], out[tag + "labels"], out[tag + "batch"] = pos, labels, batch
        out[tag + "params"] = np.array([nsample, radius, 10], np.float64)
        out[tag + "points"] = np.concatenate([c.cpu().numpy() for c in clusters]) if clusters else np.zeros(0, np.int64)
        out[tag + "offsets"] = np.concatenate([[0], np.cumsum([len(c) for c in clusters])]).astype(np.int64)
    path = os.path.join(ROOT, "tests", "golden", "me_tpk_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays; MinkowskiEngine", ME.__version__)


if __name__ == "__main__":
    main()
