#!/bin/bash
# in-tree library vs libsph3d_ab.so: step time and the in-step device time of every op family (HIP events)
cd $GRAFT_REPO_ROOT
for lib in "" "$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so"; do
  SPH3D_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=%s'%('$lib'[-14:] or 'in-tree'), d['ms_per_step'], json.dumps(d['families_ms_per_step']))"
done
