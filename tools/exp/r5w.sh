set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
VSOM_LIB=tools/exp/bin/libvsom_dev.so python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_fused.hsaco | tee gpurun_out/r5w/ab_fused.jsonl
AB_MAP=64 VSOM_LIB=tools/exp/bin/libvsom_dev.so python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_fused.hsaco | tee -a gpurun_out/r5w/ab_fused.jsonl
