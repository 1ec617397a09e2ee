import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/profiles")
import bench, train_microbench as tm
from panopticsegforlargescalepointcloud_amd.training import train_step
dev = torch.device("cuda", 0)
scene, tiles, _ = bench.build_scene(80_000 * 4, 2, 0.05, 2022)
for rep in range(2):
    model = bench.build_model(dev, 0.05)[0].train()
    data, n = tm.make_batch(scene, tiles, [0, 1, 2, 3]); data = data.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ls = []
    for it in range(6):
        train_step(model, data, opt, 1, dev, 1); ls.append(float(model.loss))
    print(os.environ.get("TAG", ""), " ".join("%.4f" % v for v in ls))
