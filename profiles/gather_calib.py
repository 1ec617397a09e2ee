"""FETCH_SIZE / WRITE_SIZE calibration for 64-byte row gathers (the access pattern of the convolution's A operand).
The guide's x2 FETCH_SIZE correction for gfx950 is calibrated on wide streaming loads (pp_triad); this script produces
launches with KNOWN HBM traffic for rocprofv3 --pmc passes (profiles/calibrate_fetch.sh):
    triad      a = b + s*c over 1 GiB arrays                              reads 2.15 GB, writes 1.07 GB   (streaming)
    gather64   out[i] = src[perm[i]], 64-byte rows, perm = random permutation of 32 M rows (2 GiB table, far beyond the
               256 MiB infinity cache): every row is fetched exactly once  reads 2.15 GB (rows) + 0.27 GB (index), writes 2.15 GB
    gather64s  the same kernel with the identity permutation                (streaming 64-byte rows)
Round 5: the same for the row widths the convolutions actually gather -- 128, 192, 256 and 384 bytes (32 / 48 / 64 / 96 fp32 channels):
    python profiles/gather_calib.py random|ident [row_bytes]    (2 GiB table whatever the width)
usage (GPU box): python profiles/gather_calib.py random|ident [row_bytes]  (one gather pattern per process: the PMC summary averages per kernel)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panopticsegforlargescalepointcloud_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    row_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n = (1 << 31) // row_bytes
    src = torch.randn(n, row_bytes // 4, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    perm = torch.randperm(n, generator=g).to(dev)
    ident = torch.arange(n, device=dev)
    a = torch.empty(1 << 28, device=dev)
    b = torch.ones(1 << 28, device=dev)
    c = torch.ones(1 << 28, device=dev)
    mode = sys.argv[1] if len(sys.argv) > 1 else "random"
    index = perm if mode == "random" else ident
    for _ in range(3):
        ops.triad(a, b, c, 2.0)
    for _ in range(3):
        ops.gather_rows(src, index)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gather_rows(src, index)
    e1.record()
    torch.cuda.synchronize()
    print("gather%d %s: %.3f ms (%.0f GB/s of rows+index+out)" % (row_bytes, mode, e0.elapsed_time(e1),
                                                                 (n * (2 * row_bytes + 8)) / e0.elapsed_time(e1) / 1e6))


if __name__ == "__main__":
    main()
