# A/B of library builds on the same box: production kernel times (stop_sweep with no stops)
R=$GRAFT_REPO_ROOT; cd $R
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for lib in ${LIBS:-liburmapx.so liburmapx_w5.so liburmapx_w6.so}; do
  echo "== $lib"
  URMAPX_LIB=$R/urmap_amd/$lib SWEEP_CHECK=${SWEEP_CHECK:-100000} python3 scripts/stop_sweep.py ${MBP:-3100} ${L:-150} ${SUB:-0.01} ${INDEL:-0.001} 1000000 0 2>&1 | grep "production\|parity"
done
rm -rf /dev/shm/urmap_idx
