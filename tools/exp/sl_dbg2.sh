# development: the integer contraction's ring kernel alone under rocprofv3 (kernel average), parts switched off (VSOM_SL_DBG)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ring in old; do for d in 0 8 16 24 32 40 4; do
  rm -rf /tmp/sld; 
  VSOM_LIB=$R/tools/exp/bin/libvsom_dev.so VSOM_SL_RING=$ring VSOM_SL_DBG=$d timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/sld -o r -- python3 $R/tools/exp/c3_search_time.py > /dev/null 2>&1
  echo -n "ring=$ring dbg=$d  "; python3 $R/tools/rocpd_summary.py /tmp/sld/r_results.db 2>/dev/null | grep "sl_gemm_i8_ring" | head -1 | awk '{print $1, $3}'
done; done
