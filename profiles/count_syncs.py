import sys, collections, traceback
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--points", "1400000", "--grid", "3"]
sys.path.insert(0, "/root/repo")
import torch
cnt = collections.Counter()
def _site():
    """innermost frame inside the package (not this script, not torch)"""
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "panopticsegforlargescalepointcloud_amd" in fr.filename:
            return (fr.filename.split("/")[-1], fr.lineno, fr.line.strip()[:70])
    fr = traceback.extract_stack()[-3]
    return (fr.filename.split("/")[-1], fr.lineno, (fr.line or "").strip()[:70])
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            cnt[(name,) + _site()] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("item", "tolist", "cpu", "numpy"):
    wrap(n)
_orig_bool = torch.Tensor.__bool__
def _b(self):
    if self.is_cuda:
        cnt[("bool",) + _site()] += 1
    return _orig_bool(self)
torch.Tensor.__bool__ = _b
_orig_int = torch.Tensor.__int__
def _i(self):
    if self.is_cuda:
        cnt[("int",) + _site()] += 1
    return _orig_int(self)
torch.Tensor.__int__ = _i
import runpy
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
tot = 0
for k, v in cnt.most_common(45):
    print(v / 5.0, k)
    tot += v
print("total per step", sum(cnt.values()) / 5.0)
