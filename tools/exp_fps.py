"""FPS chain times at the plans' shapes: pruned kernel vs the plain register kernel (SPH3D_FPS_PRUNE=0 in a child process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from sph3d_gcn_amd import tf_sample, _lib
    from sph3d_gcn_amd.harness import synth
    dev = torch.device("cuda:0"); _lib.lib()
    def timeit(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for kind, B, N, m in (("s3dis", 16, 8192, 2048), ("s3dis", 16, 2048, 768), ("modelnet", 32, 10000, 2500), ("modelnet", 32, 2500, 625),
                          ("s3dis", 64, 2048, 512), ("s3dis", 16, 4096, 1024)):
        if kind == "s3dis":
            xyz = synth.s3dis_batch(1000, B, N)[0][:, :, :3].copy()
        else:
            xyz = synth.modelnet_batch(100, B, N)[:, :, :3].copy()
        x = torch.from_numpy(xyz).to(dev)
        if N == 2048 and B == 16:      # the real level-1 input: the FPS samples of level 0
            big = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0][:, :, :3].copy()).to(dev)
            idx = tf_sample.farthest_point_sample(2048, big).long()
            x = torch.gather(big, 1, idx.unsqueeze(2).expand(-1, -1, 3)).contiguous()
        t = timeit(lambda: tf_sample.farthest_point_sample(m, x))
        print("prune=%s %-8s B%3d N%6d -> %5d : %8.1f us  %.3f us/round" % (os.environ.get("SPH3D_FPS_PRUNE", "default"), kind, B, N, m, t * 1e3, t * 1e3 / (m - 1)))
else:
    for v in ("2049", "1025", "0"):
        env = dict(os.environ, SPH3D_FPS_PRUNE=v)
        print("---- SPH3D_FPS_PRUNE=%s" % v, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
