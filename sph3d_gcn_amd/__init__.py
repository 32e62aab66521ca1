"""sph3d_gcn_amd — MI355X-native implementation of the SPH3D-GCN ``tf_ops`` hot path.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/sph3d.h) -> libsph3d.so
  _lib.py          ctypes binding of the C ABI (no fallback: raises if the library is missing)
  tf_nnquery.py tf_buildkernel.py tf_conv3d.py tf_pool3d.py tf_unpool3d.py tf_sample.py
                   op-level modules with the reference's names and signatures (tf_ops/*/tf_*.py)
  tf_gemm.py       the pointwise 1x1 feature GEMM (tf.matmul in the reference)
  sph3gcn_util.py  the glue with the reference's public signatures (utils/sph3gcn_util.py)
  harness/         torch re-statement of the model graphs' call pattern, used by bench/smoke only
"""
from . import _lib  # noqa: F401
from . import tf_nnquery, tf_buildkernel, tf_conv3d, tf_pool3d, tf_unpool3d, tf_sample, tf_gemm, tf_norm  # noqa: F401
from . import sph3gcn_util  # noqa: F401

__all__ = ["tf_nnquery", "tf_buildkernel", "tf_conv3d", "tf_pool3d", "tf_unpool3d", "tf_sample", "tf_gemm",
           "sph3gcn_util"]
