set -e
for sp in "" "12,43" "0,57" "50,0" "40,11"; do
  echo "794 split=$sp"
  VSOM_UPD_SPLIT=$sp python bench.py --no-cpu --dim 794 --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
done
echo "785 (dense iid data, 56x14 + pad)"
python bench.py --no-cpu --dim 785 --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
