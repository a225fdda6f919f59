import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import torch
from discorpy_amd import _ffi as F, configs
from discorpy_amd.post import postprocessing as pp
F.require_device()
c = configs.cfg2(); s = 4096.0/256
xc, yc = c["xcenter"]/s, c["ycenter"]/s
fact = [v*s**k for k, v in enumerate(c["list_fact"])]
t = torch.rand((256,256), device="cuda"); o = torch.empty_like(t)
for i in range(100): pp.unwarp_image_backward(t, xc, yc, fact, out=o)
pr = cProfile.Profile(); pr.enable()
for i in range(3000): pp.unwarp_image_backward(t, xc, yc, fact, out=o)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
