#!/bin/bash
# exact memory-side read bytes from the request-size counters (no x2 guess): RDREQ (all), RDREQ_32B, BUBBLE (128-byte requests if the counter exists)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
export PYTHONPATH=$PWD TMPDIR=/tmp
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i -B1 -A4 "TCC_BUBBLE\|TCC_EA0_RDREQ_128\|TCC_REQ\b" | head -40) > $O/avail.txt
cat $O/avail.txt | head -30
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/f_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc-calib --prewarm-s 0.1) > $O/f_$n.log 2>&1
  python tools/rocpd_summary.py $O/f_$n/*/*.db $O/fwd_$n.md | grep -E "tsw_kernel|elementwise_kernel<4" | grep -v "^| void.*| [0-9]* | [0-9.]* | [0-9.]* |" | cut -c1-170
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/b_$n -- python $GRAFT_REPO_ROOT/tools/bench_backward.py --batch 64 --steps 3) > $O/b_$n.log 2>&1
  python tools/rocpd_summary.py $O/b_$n/*/*.db $O/bwd_$n.md | grep -E "bwd_final|tsw_kernel" | grep -v "^| void.*| [0-9]* | [0-9.]* | [0-9.]* |" | cut -c1-170
  rm -rf $O/f_$n $O/b_$n
done
