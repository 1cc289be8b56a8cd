#!/bin/bash
# round 6 (GPU box): is the box-to-box spread of the transfer-inclusive rate (31-58 M poses/s at 10 k in round 5) a matter of WHERE the feeding
# thread runs?  The same bench (no host baselines, so bench.py sets no OpenMP binding) with the process confined to the GPU's NUMA node, to
# the other node, and unconfined; the line's transfer_inclusive.host_link says where the device hangs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06; out=gpurun_out/r06/xfer_numa.txt; : > $out
for n in /sys/devices/system/node/node[0-9]*; do echo "$(basename $n): cpus $(cat $n/cpulist)" >> $out; done
one() { # label, prefix...
  label=$1; shift
  for i in 1 2; do
    "$@" python bench.py --no-cpu-baseline --pmc off --no-overlap --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); t=d['transfer_inclusive']
print('$label', 'one_stream %.1f M' % (t['one_stream']['value']/1e6), 'two_streams %.1f M' % (t['two_streams']['value']/1e6), 'device on numa node', t['host_link'].get('numa_node'), t['host_link'].get('current_link_speed'), 'x' + str(t['host_link'].get('current_link_width')), '| resident %.1f M' % (d['value']/1e6))" >> $out
  done
}
one "unconfined       "
for n in /sys/devices/system/node/node[0-9]*; do
  c=$(cat $n/cpulist)
  one "taskset $(basename $n)" taskset -c $c
  first=$(echo $c | sed 's/[-,].*//')
  one "one cpu of $(basename $n)" taskset -c $first
done
cat $out
