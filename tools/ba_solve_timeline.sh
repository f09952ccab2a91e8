# in-kernel phase split of the pose solve at the window sizes of the bench and of a real frontend update (tools/ba_solve_timeline.py --build first, here)
for cfg in "NF=8" "NF=13 RAD=8 HT=30 WD=101" "NF=22 RAD=8 HT=30 WD=101" "NF=26 RAD=8 HT=30 WD=101" "NF=30 RAD=8 HT=30 WD=101"; do
  echo "== $cfg"; env $cfg python tools/ba_solve_timeline.py 2>&1 | grep -v amdgpu.ids
done
