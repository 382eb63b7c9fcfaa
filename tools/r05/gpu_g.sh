cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
for v in product nobar noloop; do
  if [ $v = product ]; then unset CSPN_AMD_LIB; else export CSPN_AMD_LIB=$PWD/cspn_amd/abl/libcspn_$v.so; fi
  timeout 300 python tools/bench_backward.py --batch 64 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v',d['ms_per_call'])"
done
