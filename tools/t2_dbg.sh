export TMPDIR=/tmp
for d in 0 1 2 3; do echo "== dbg $d"; SPH3D_T2_DBG=$d UCAPS=112 timeout 100 python tools/exp_tile2.py 2>&1 | grep -E "order 1|C=" | head -4; done
echo "== ucap 96"; UCAPS=96 timeout 100 python tools/exp_tile2.py 2>&1 | grep -E "order 1|C=" | head -4
