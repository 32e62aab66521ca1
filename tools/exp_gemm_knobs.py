"""tools/exp_gemm.py under the tile-choice knobs of csrc/gemm.hip (SPH3D_TN_WGS: workgroups the weight gradient's split-K aims at;
SPH3D_GEMM_MINTILES: tiles a product needs before it takes the larger tile), each setting in a child process."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for name, vals in (("SPH3D_TN_WGS", ("256", "384", "512", "768", "1024")), ("SPH3D_GEMM_MINTILES", ("192", "256", "384", "512", "768"))):
    for v in vals:
        print("---- %s=%s" % (name, v), flush=True)
        subprocess.run([sys.executable, os.path.join(here, "exp_gemm.py")], env=dict(os.environ, **{name: v}), check=False)
