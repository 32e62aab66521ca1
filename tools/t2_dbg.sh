export TMPDIR=/tmp
python tools/exp_fwd_variants.py 2>&1 | tail -1
for v in w8s4 w8s8 w4s8 w16s4; do SPH3D_LIB=$PWD/sph3d_gcn_amd/csrc/libsph3d_$v.so python tools/exp_fwd_variants.py 2>&1 | tail -1; done
python tools/exp_fwd_variants.py 2>&1 | tail -1
