"""Bench / smoke harness: synthetic inputs, the SPH3D_s3dis call pattern on s3g_util, data-parallel step.
Not part of the drop-in surface."""
