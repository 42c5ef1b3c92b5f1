# development: what the integer contraction's ring kernel spends its time on (timing only, WRONG results):
# VSOM_SL_DBG 1 = one K chunk only, 2 = no epilogue, 4 = no G stores; both tile shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ring in old two; do for d in 0 1 2 3 4; do
  echo -n "ring=$ring dbg=$d  "
  VSOM_LIB=$PWD/tools/exp/bin/libvsom_dev.so VSOM_SL_RING=$ring VSOM_SL_DBG=$d timeout -k 10 120 python tools/exp/c3_search_time.py 2>/dev/null | tail -1
done; done
