for v in grp4 grp16 grp32; do
for n in 8 24; do
echo -n "$v "; TMX_VI_GROUP=1 TMX_LIBRARY=$PWD/build/abl/$v.so python3 tools/rank_share_timing.py $n 2>&1 | grep "^ranks" | cut -c1-200
done; done
