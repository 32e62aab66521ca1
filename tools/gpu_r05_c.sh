#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_full_configs.py -m gpu -q -x -k "fps or sampl or full or config" 2>&1 | tail -3 | tee $O/r05c_tests.log
SPH3D_FPS_PRUNE=0 timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py -m gpu -q -x -k "fps" 2>&1 | tail -3 | tee -a $O/r05c_tests.log
timeout 300 python tools/exp_fps.py > $O/r05c_fps.log 2>&1; grep -v amdgpu $O/r05c_fps.log
python - <<PY
import torch, sys
sys.path.insert(0, ".")
from sph3d_gcn_amd import tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0")
big = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0][:, :, :3].copy()).to(dev)
cur = big
for m in (2048, 768, 384, 128):
    idx = tf_sample.farthest_point_sample(m, cur); nxt = torch.gather(cur, 1, idx.long().unsqueeze(2).expand(-1, -1, 3)).contiguous()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): tf_sample.farthest_point_sample(m, cur)
    e1.record(); torch.cuda.synchronize()
    print("chain level %d -> %d: %.1f us, %.3f us/round" % (cur.shape[1], m, e0.elapsed_time(e1) * 200, e0.elapsed_time(e1) * 200 / (m - 1)))
    cur = nxt
PY
