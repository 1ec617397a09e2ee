import sys, collections, traceback
sys.argv = ["bench.py", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--points", "1400000", "--grid", "3"]
sys.path.insert(0, "/root/repo")
import torch
cnt = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda:
            fr = traceback.extract_stack(limit=3)[0]
            cnt[(name, fr.filename.split("/")[-1], fr.lineno)] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ("item", "tolist", "cpu", "numpy"):
    wrap(n)
_orig_bool = torch.Tensor.__bool__
def _b(self):
    if self.is_cuda:
        fr = traceback.extract_stack(limit=2)[0]
        cnt[("bool", fr.filename.split("/")[-1], fr.lineno)] += 1
    return _orig_bool(self)
torch.Tensor.__bool__ = _b
_orig_int = torch.Tensor.__int__
def _i(self):
    if self.is_cuda:
        fr = traceback.extract_stack(limit=2)[0]
        cnt[("int", fr.filename.split("/")[-1], fr.lineno)] += 1
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
