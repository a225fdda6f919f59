import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import noise
from discorpy_amd.util import utility as util
from discorpy_amd.post import postprocessing as pp
from discorpy_amd import _ffi as F
from oracle import oracle as orc
orc.build()
FACT5 = [1.00227490554, -9.3601153805625e-06, 8.78436609375e-09, -4.79328802218628e-12, 7.714082828693389e-16]
rgb = noise(11, (1000,1536,3))*255.0
xc,yc=700.3,480.9
for order, blend, ob in ((1,"scipy",orc.BLEND_SCIPY),(0,None,orc.BLEND_SCIPY)):
    got = util.unwarp_color_image_backward(rgb, xc, yc, FACT5, order=order, blend=blend)
    print(F.last_kernel())
    want = np.stack([orc.unwarp_image_backward(np.ascontiguousarray(rgb[:,:,c]), xc, yc, FACT5, order=order, poly=orc.POLY_KERNEL, blend=ob) for c in range(3)], axis=2)
    k1 = np.stack([pp.unwarp_image_backward(np.ascontiguousarray(rgb[:,:,c]), xc, yc, FACT5, order=order, blend=blend) for c in range(3)], axis=2)
    d = np.argwhere(got != want)
    print(order, blend, len(d), "k1 vs oracle diffs", int((k1!=want).sum()))
    if len(d):
        print("rows", np.unique(d[:,0])[:20], "cols range", d[:,1].min(), d[:,1].max(), "chan", np.unique(d[:,2]))
        for (y,x,c) in d[:6]:
            print(y,x,c, got[y,x,c], want[y,x,c], k1[y,x,c])
