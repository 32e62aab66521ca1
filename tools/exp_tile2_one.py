"""one shape of the tile2 forward, few launches (for rocprofv3 --pmc)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, _plan, tf_conv3d, tf_nnquery
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0")
B, C, F = 16, int(os.environ.get("C", 128)), 33
ucap = int(os.environ.get("UCAP", 252))
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
idx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, 64, [8, 2, 2], with_transpose=False)
_plan.register_geometry(filt, xyz, xyz)
chdr, rec, ulist, _ = _plan.forward_plan2(idx, cnt, filt, F, ucap, True)
x = torch.randn(B, 8192, C, device=dev)
w = torch.randn(F, C, 2, device=dev)
out = torch.empty(B, 8192, 2 * C, device=dev)
l = _lib.lib()
for _ in range(3):
    _lib.check(l.sph3d_depthwise_conv3d_tiled2(B, 8192, 8192, F, C, 2, ucap, _lib.ptr(chdr), _lib.ptr(rec), _lib.ptr(ulist),
                                               _lib.ptr(x), _lib.ptr(w), _lib.ptr(out), _lib.stream_ptr()))
    tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
torch.cuda.synchronize()
