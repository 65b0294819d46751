# overlap cost table with two builds of the library alternating on one box
cd $GRAFT_REPO_ROOT
L=videovector_amd/lib
cp $L/libvideovec.so $L/libvideovec_new.so
for r in 1 2; do for b in prev new; do
  cp $L/libvideovec_$b.so $L/libvideovec.so
  echo "== $b"; timeout 600 python3 tools/lab/overlap_cost.py 600 2>&1 | grep -E "ms/step" | grep -E "^sync|^overlap  |overlap delay"
done; done
cp $L/libvideovec_new.so $L/libvideovec.so
timeout 300 python -m pytest tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -1
